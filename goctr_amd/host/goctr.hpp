// goctr.hpp -- C++ host mirror of go-ctr's operator surface above the C-ABI (include/goctr.h).
//
// The reference is compiled Go and the Go toolchain is absent from the build image, so this header restates
// the Go-side contracts of the hot path (recommend/rcmd.go:56-97,132-137; model/model.go:16-33,215-242;
// model/din/din.go:21-52,62-211; model/youtube/dnn.go; model/mlp/mlp.go:15-65;
// feature/embedding/wordemb.go:9-32) with the same names, argument order and error behaviour.  It moves
// buffers only: every number comes out of libgoctr_hip.so.  Errors surface as std::runtime_error carrying
// goctr_last_error() (the Go adapters log.Fatalf / return error at the same places).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/goctr.h"

namespace goctr {

inline void check(int rc) {
  if (rc != 0) throw std::runtime_error(goctr_last_error());
}
// bind the process to GPU 0 unless something (goctr_init / goctr_init_devices) already bound it
inline void ensure_init() {
  char name[8];
  if (goctr_device_info(name, sizeof name, nullptr, nullptr) != 0) check(goctr_init(0));
}
// one process, n ranks (SURVEY 8(b) goctr_init(n_devices, ids)): engine k on HIP device ids[k]; a repeated id = logical ranks on one device.
// Training calls then take `devices = ids.size()` (model::Train below): recommend.Train reaches n GPUs unchanged.
inline void InitDevices(const std::vector<int>& ids) { check(goctr_init_devices((int)ids.size(), ids.data())); }

namespace recommend {
// recommend/rcmd.go:19-28
constexpr int ItemEmbDim = 16, ItemEmbWindow = 5, UserBehaviorLen = 10;

struct SampleInfo {  // rcmd.go:132-137
  std::array<int, 2> UserProfileRange{}, UserBehaviorRange{}, ItemFeatureRange{}, CtxFeatureRange{};
  std::array<int, 8> ranges() const {
    return {UserProfileRange[0], UserProfileRange[1], UserBehaviorRange[0], UserBehaviorRange[1],
            ItemFeatureRange[0], ItemFeatureRange[1], CtxFeatureRange[0], CtxFeatureRange[1]};
  }
  static SampleInfo FromDims(int U, int T, int D, int C) {  // rcmd.go:401-422
    SampleInfo s;
    s.UserProfileRange = {0, U};
    s.UserBehaviorRange = {U, U + T * D};
    s.ItemFeatureRange = {U + T * D, U + T * D + D};
    s.CtxFeatureRange = {U + T * D + D, U + T * D + D + C};
    return s;
  }
};

struct TrainSample {  // rcmd.go:56-63
  std::vector<float> X, Y;
  int Rows = 0, XCols = 0;
  SampleInfo Info;
};

struct PredictAbstract {  // rcmd.go:87-89
  virtual ~PredictAbstract() = default;
  virtual std::vector<float> Predict(const float* X, int64_t rows, int xcols) = 0;
};
struct Fitter {  // rcmd.go:95-97
  virtual ~Fitter() = default;
  virtual std::shared_ptr<PredictAbstract> Fit(const TrainSample& s) = 0;
};
}  // namespace recommend

namespace model {
constexpr int mlp0_1 = 200, mlp1_2 = 80;  // din.go:17-18

class CtrNet {  // the device twin of model.Model (model.go:16-25)
 public:
  CtrNet(int kind, int U, int T, int D, int iD, int C, int att = GOCTR_ATT_COSINE) : U(U), T(T), D(D), C(C) {
    if (kind == GOCTR_DIN && D != iD)  // din.go:176-178
      throw std::invalid_argument("uBehaviorDim != iFeatureDim");
    ensure_init();
    goctr_ctr_cfg cfg{kind, att, U, T, D, C, mlp0_1, mlp1_2};
    check(goctr_model_create(&cfg, &h_));
  }
  ~CtrNet() { goctr_model_destroy(h_); }
  CtrNet(const CtrNet&) = delete;
  CtrNet& operator=(const CtrNet&) = delete;
  goctr_model* Vm() const { return h_; }
  void SetWeights(int tensor, const std::vector<float>& w) { check(goctr_model_set_weights(h_, tensor, w.data(), w.size())); }
  // checkpoint / resume: Adam moments (which = 0 first, 1 second) and the step counter
  void SetMoments(int tensor, int which, const std::vector<float>& w) { check(goctr_model_set_moments(h_, tensor, which, w.data(), w.size())); }
  void GetMoments(int tensor, int which, std::vector<float>& w) const { check(goctr_model_get_moments(h_, tensor, which, w.data(), w.size())); }
  uint32_t Step() const { uint32_t s = 0; check(goctr_model_get_step(h_, &s)); return s; }
  void SetStep(uint32_t s) { check(goctr_model_set_step(h_, s)); }
  // EXTENSION (no reference counterpart): lr > 0 trains the embedding table too (SGD scatter-add)
  void SetEmbeddingTraining(double lr) { check(goctr_model_set_embedding_training(h_, lr)); }
  std::vector<float> GetWeights(int tensor) const {
    const int I = U + 2 * D + C;
    const size_t n = tensor == GOCTR_W0 ? (size_t)I * mlp0_1 : tensor == GOCTR_W1 ? (size_t)mlp0_1 * mlp1_2
                   : tensor == GOCTR_W2 ? (size_t)mlp1_2 : (size_t)T;
    std::vector<float> w(n);
    check(goctr_model_get_weights(h_, tensor, w.data(), n));
    return w;
  }
  // G.Gaussian(0,1) init (din.go:187-191)
  void InitGaussian(uint64_t seed) {
    std::mt19937_64 g(seed);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int t : {GOCTR_W0, GOCTR_W1, GOCTR_W2}) {
      auto w = GetWeights(t);
      for (auto& v : w) v = nd(g);
      SetWeights(t, w);
    }
  }
  int U, T, D, C;

 private:
  goctr_model* h_ = nullptr;
};

// model.Train (model.go:27-33); returns the per-epoch costs the Go version logs (model.go:205)
inline std::vector<float> Train(int /*uProfileDim*/, int /*uBehaviorSize*/, int /*uBehaviorDim*/, int /*iFeatureDim*/,
                                int /*cFeatureDim*/, int numExamples, int batchSize, int epochs, int earlyStop,
                                const recommend::SampleInfo& si, const float* inputs, int xcols, const float* targets,
                                CtrNet& m, int devices = 0) {
  goctr_train_cfg cfg;
  goctr_train_cfg_default(&cfg);
  cfg.batch = batchSize; cfg.epochs = epochs; cfg.early_stop = earlyStop;
  cfg.devices = devices;        // n of InitDevices: batchSize stays the GLOBAL batch, sharded over the n ranks inside this one call
  auto r = si.ranges();
  std::vector<float> costs((size_t)std::max(epochs, 1));
  int ran = 0;
  check(goctr_train_dense(m.Vm(), inputs, targets, numExamples, xcols, r.data(), &cfg, costs.data(), &ran));
  costs.resize((size_t)ran);
  return costs;
}

// model.Predict (model.go:242)
inline std::vector<float> Predict(CtrNet& m, int numExamples, int batchSize, const recommend::SampleInfo& si,
                                  const float* inputs, int xcols) {
  std::vector<float> y((size_t)numExamples);
  auto r = si.ranges();
  check(goctr_predict_dense(m.Vm(), inputs, numExamples, xcols, r.data(), batchSize, y.data()));
  return y;
}
}  // namespace model

namespace recommend {
// rcmd.go:65-71, 118-121
struct Sample { int UserId = 0, ItemId = 0; float Label = 0; int64_t Timestamp = 0; };
struct ItemScore { int ItemId = 0; float Score = 0; };

// What GetSampleVector reads per key (rcmd.go:462-536), resident in HBM: user / item feature tables (rows = dense user /
// item index), the behaviour cache CSR (feature/ubcache/cache.go) and the item-embedding table.  Serving side of the
// drop-in: BatchPredict / Rank (rcmd.go:277-337, 248-275) hand the device KEYS, not 281-wide rows.
class RecSys {
 public:
  RecSys(const std::vector<int64_t>& ub_off, const std::vector<int32_t>& ub_items, const std::vector<int64_t>& ub_ts,
         const std::vector<float>& user_table, int U, const std::vector<float>& item_table, int C,
         const std::vector<float>& item_emb, int D) {
    ensure_init();
    const int64_t n_users = (int64_t)ub_off.size() - 1, n_items = C ? (int64_t)item_table.size() / C : 0;
    check(goctr_ubcache_create(n_users, ub_off.data(), ub_items.data(), ub_ts.data(), &ub_));
    check(goctr_emb_create((int64_t)item_emb.size() / D, D, item_emb.data(), &emb_));
    check(goctr_recsys_create(ub_, emb_, user_table.data(), n_users, U, item_table.data(), n_items, C, &h_));
  }
  ~RecSys() { goctr_recsys_destroy(h_); goctr_emb_destroy(emb_); goctr_ubcache_destroy(ub_); }
  RecSys(const RecSys&) = delete;
  RecSys& operator=(const RecSys&) = delete;
  goctr_recsys* handle() const { return h_; }
  goctr_emb* embedding() const { return emb_; }

 private:
  goctr_ubcache* ub_ = nullptr; goctr_emb* emb_ = nullptr; goctr_recsys* h_ = nullptr;
};

// rcmd.go:277-337.  Throws when the first key fails (rcmd.go:293-296) and -- the reference's named-result quirk -- when
// the last one does (rcmd.go:291,325-336); keys failing in between score as the all-zero row (rcmd.go:299-302).
inline std::vector<float> BatchPredict(model::CtrNet& net, RecSys& rs, const std::vector<Sample>& keys, int predBatch = 4096) {
  const int64_t n = (int64_t)keys.size();
  std::vector<int32_t> u((size_t)n), it((size_t)n);
  std::vector<int64_t> ts((size_t)n);
  for (int64_t i = 0; i < n; ++i) { u[i] = keys[i].UserId; it[i] = keys[i].ItemId; ts[i] = keys[i].Timestamp; }
  std::vector<float> y((size_t)n);
  std::vector<uint8_t> failed((size_t)n);
  check(goctr_batch_predict(net.Vm(), rs.handle(), u.data(), it.data(), ts.data(), n, predBatch, y.data(), failed.data(), nullptr));
  if (n && failed[(size_t)n - 1]) throw std::runtime_error("get sample vector error: last key has no features");
  return y;
}

// rcmd.go:248-275
inline std::vector<ItemScore> Rank(model::CtrNet& net, RecSys& rs, int userId, const std::vector<int>& itemIds, int64_t now,
                                   int predBatch = 4096) {
  // one user, one timestamp: goctr_rank takes them as scalars (no per-key arrays are built or copied)
  const int64_t n = (int64_t)itemIds.size();
  std::vector<float> y((size_t)n);
  std::vector<uint8_t> failed((size_t)n);
  static_assert(sizeof(int) == sizeof(int32_t), "item ids are int32");
  check(goctr_rank(net.Vm(), rs.handle(), userId, reinterpret_cast<const int32_t*>(itemIds.data()), n, now, predBatch, y.data(),
                   failed.data(), nullptr));
  if (n && failed[(size_t)n - 1]) throw std::runtime_error("get sample vector error: last key has no features");   // rcmd.go:258-260
  std::vector<ItemScore> out(itemIds.size());
  for (size_t i = 0; i < itemIds.size(); ++i) out[i] = ItemScore{itemIds[i], y[i]};
  return out;
}
}  // namespace recommend

namespace din {
struct DinNet : model::CtrNet {
  DinNet(int U, int T, int D, int iD, int C) : CtrNet(GOCTR_DIN, U, T, D, iD, C) {}
};
}  // namespace din
namespace youtube {
struct YoutubeDnn : model::CtrNet {
  YoutubeDnn(int U, int T, int D, int iD, int C) : CtrNet(GOCTR_YOUTUBE, U, T, D, iD, C) {}
};
}  // namespace youtube

namespace mlp {
// nn.MLPClassifier behind model/mlp's wrappers (multilayer_perceptron.go:81-125, mlp.go:15-65)
class MLPClassifier {
 public:
  std::vector<int> HiddenLayerSizes{100};
  int Activation = GOCTR_ACT_RELU, Solver = GOCTR_SOLVER_ADAM;
  double Alpha = 1e-4, LearningRateInit = 1e-3;
  int BatchSize = 200, MaxIter = 200;
  uint64_t RandomState = 1;
  std::vector<double> LossCurve;
  ~MLPClassifier() { goctr_mlp_destroy(h_); }
  void Fit(const float* X, const float* Y, int64_t rows, int xcols) {
    ensure_init();
    goctr_mlp_cfg cfg;
    goctr_mlp_cfg_default(&cfg);
    std::vector<int> units{xcols};
    units.insert(units.end(), HiddenLayerSizes.begin(), HiddenLayerSizes.end());
    units.push_back(1);
    cfg.n_layers = (int)units.size();
    for (size_t i = 0; i < units.size(); ++i) cfg.units[i] = units[i];
    cfg.activation = Activation; cfg.solver = Solver; cfg.alpha = Alpha; cfg.lr_init = LearningRateInit;
    cfg.batch = BatchSize; cfg.max_iter = MaxIter;
    goctr_mlp_destroy(h_);
    check(goctr_mlp_create(&cfg, &h_));
    // initialize (basemlp64.go:466-475): U[0,1) * sqrt(f / (fanIn + fanOut)), one-sided (quirk Q8)
    std::mt19937_64 g(RandomState);
    std::uniform_real_distribution<double> ud(0.0, 1.0);
    std::vector<double> theta;
    for (size_t i = 0; i + 1 < units.size(); ++i) {
      const double bound = std::sqrt((Activation == GOCTR_ACT_LOGISTIC ? 2.0 : 6.0) / (units[i] + units[i + 1]));
      for (int k = 0; k < (1 + units[i]) * units[i + 1]; ++k) theta.push_back(ud(g) * bound);
    }
    check(goctr_mlp_set_params(h_, theta.data(), theta.size()));
    const int64_t use = rows / BatchSize * BatchSize;
    LossCurve.assign((size_t)MaxIter, 0.0);
    int iters = 0;
    check(goctr_mlp_fit(h_, X, Y, use, nullptr, LossCurve.data(), &iters));
    LossCurve.resize((size_t)iters);
  }
  std::vector<float> Predict(const float* X, int64_t rows) {
    std::vector<float> y((size_t)rows);
    check(goctr_mlp_predict(h_, X, rows, y.data()));
    return y;
  }

 private:
  goctr_mlp* h_ = nullptr;
};
}  // namespace mlp

namespace embedding {
// TrainEmbedding's device half (wordemb.go:9-32): the caller owns the dictionary and passes counts + id doc
struct Model {
  int64_t V = 0; int dim = 0;
  std::vector<float> vectors;  // GenEmbeddingMap32 rows (word2vec.go:298-324)
};
inline Model TrainEmbedding(const std::vector<int64_t>& counts, const std::vector<int32_t>& doc, int64_t corpus_len,
                            int window, int dim, int iter, uint64_t seed = 1, int devices = 0) {
  ensure_init();
  goctr_w2v_cfg cfg;
  goctr_w2v_cfg_default(&cfg);
  cfg.dim = dim; cfg.window = window; cfg.devices = devices;   // devices = n of InitDevices: every pass data-parallel in one call
  goctr_w2v* h = nullptr;
  check(goctr_w2v_create(&cfg, (int64_t)counts.size(), counts.data(), &h));
  std::mt19937_64 g(seed);
  std::uniform_real_distribution<double> ud(0.0, 1.0);
  std::vector<double> param(counts.size() * (size_t)dim);
  for (auto& v : param) v = (ud(g) - 0.5) / dim;  // word2vec.go:103-111
  check(goctr_w2v_set_param(h, param.data()));
  double lr = cfg.init_lr;
  for (int it = 0; it < iter; ++it) check(goctr_w2v_train(h, doc.data(), (int64_t)doc.size(), corpus_len, nullptr, &lr));
  Model m;
  m.V = (int64_t)counts.size(); m.dim = dim;
  m.vectors.resize(counts.size() * (size_t)dim);
  check(goctr_w2v_export_f32(h, m.vectors.data()));
  goctr_w2v_destroy(h);
  return m;
}

// The same with the corpus load on the device (memory.go:76-102, dictionary.go:70-81, memory.go:53-62): the caller hands
// over the ItemSeqGenerator batches as int64 item ids; `ids` of the result is Dictionary.id2word.
struct IdModel : Model { std::vector<int64_t> ids; };
inline IdModel TrainEmbeddingIds(const std::vector<std::vector<int64_t>>& batches, int window, int dim, int iter,
                                 int64_t min_count = 5, int64_t max_count = -1, double subsample = 1e-3, uint64_t seed = 1,
                                 int devices = 0) {
  ensure_init();
  int64_t cap = 0;
  for (const auto& b : batches) cap += (int64_t)b.size();
  goctr_corpus* c = nullptr;
  check(goctr_corpus_create(std::max<int64_t>(cap, 1), &c));
  for (const auto& b : batches) check(goctr_corpus_append(c, b.data(), (int64_t)b.size()));
  check(goctr_corpus_build(c, min_count, max_count));
  int64_t n_words = 0, V = 0;
  check(goctr_corpus_info(c, &n_words, &V, nullptr));
  goctr_w2v_cfg cfg;
  goctr_w2v_cfg_default(&cfg);
  cfg.dim = dim; cfg.window = window; cfg.devices = devices;
  goctr_w2v* h = nullptr;
  check(goctr_w2v_create_from_corpus(&cfg, c, &h));
  std::mt19937_64 g(seed);
  std::uniform_real_distribution<double> ud(0.0, 1.0);
  std::vector<double> param((size_t)V * (size_t)dim);
  for (auto& v : param) v = (ud(g) - 0.5) / dim;
  check(goctr_w2v_set_param(h, param.data()));
  double lr = cfg.init_lr;
  for (int it = 0; it < iter; ++it) {
    check(goctr_w2v_use_corpus(h, c, subsample, seed + (uint64_t)it));
    check(goctr_w2v_train_resident(h, n_words, &lr));
  }
  IdModel m;
  m.V = V; m.dim = dim;
  m.ids.resize((size_t)V);
  check(goctr_corpus_get_dictionary(c, m.ids.data(), nullptr));
  m.vectors.resize((size_t)V * (size_t)dim);
  check(goctr_w2v_export_f32(h, m.vectors.data()));
  goctr_w2v_destroy(h);
  goctr_corpus_destroy(c);
  return m;
}
}  // namespace embedding

namespace search {
// feature/embedding/search/search.go: Neighbor :27-31, Searcher :52-63, SearchInternal :65-83, SearchVector :85-90
struct Neighbor { std::string Word; unsigned Rank = 0; double Similarity = 0; };
using Neighbors = std::vector<Neighbor>;
class Searcher {
 public:
  Searcher(std::vector<std::string> words, const std::vector<double>& vectors, int dim) : words_(std::move(words)), dim_(dim) {
    if (words_.empty() || vectors.size() != words_.size() * (size_t)dim) throw std::runtime_error("embeddings do not validate");
    vec_ = vectors;
    ensure_init();
    check(goctr_searcher_create(vec_.data(), (int64_t)words_.size(), dim, &h_));
  }
  ~Searcher() { if (h_) goctr_searcher_destroy(h_); }
  Searcher(const Searcher&) = delete;
  Searcher& operator=(const Searcher&) = delete;
  Neighbors SearchVector(const std::vector<double>& query, int k) { return search(query.data(), k, -1); }
  Neighbors SearchInternal(const std::string& word, int k) {
    for (size_t i = 0; i < words_.size(); ++i)
      if (words_[i] == word) return search(vec_.data() + i * (size_t)dim_, k, (int64_t)i);
    throw std::runtime_error(word + " is not found in searcher");   // search.go:73-75
  }

 private:
  Neighbors search(const double* q, int k, int64_t ignore) {
    std::vector<int64_t> idx(k);
    std::vector<double> sim(k);
    int count = 0;
    check(goctr_searcher_search(h_, q, 1, k, ignore >= 0 ? &ignore : nullptr, idx.data(), sim.data(), &count));
    Neighbors out(count);
    for (int r = 0; r < count; ++r)
      if (idx[r] >= 0) out[r] = Neighbor{words_[(size_t)idx[r]], (unsigned)r + 1, sim[r]};
    return out;
  }
  std::vector<std::string> words_;
  std::vector<double> vec_;
  int dim_;
  goctr_searcher* h_ = nullptr;
};
}  // namespace search

}  // namespace goctr
