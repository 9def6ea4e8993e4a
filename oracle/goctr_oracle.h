/*
 * goctr_oracle.h -- CPU restatement ("oracle") of auxten/go-ctr's CTR hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / the timed CPU baseline.  The
 * product (goctr_amd/csrc -> libgoctr_hip.so) never links or calls it.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it restates.  The reference is Go (+ un-vendored gorgonia /
 * gonum); there is no Go toolchain in the build image, so the reference itself
 * cannot be executed here.  Pinning status per sub-path:
 *   - op-level (cosine / euclid / prelu / bce / mse / rms / roc-auc): PINNED by the
 *     reference's own literal known-answer tests, transcribed as data into
 *     tests/golden/ref_kats.json.
 *   - item2vec primitives (LCG stream, sigmoid table): PINNED by values derivable from
 *     the reference constants (modelutil.go:21-29, sigmoid_table.go:28-45).
 *   - full DIN / YouTube forward-backward-Adam step, sklearn-port MLP training,
 *     item2vec training: "parity unpinned" -- the reference holds no golden vector
 *     for them (model_test.go asserts AUC>0.5 only; the MLP KAT dataset lives in an
 *     un-vendored module).  They are cross-checked against independent
 *     implementations (torch autograd on CPU, scikit-learn's MLP loss/grad) in
 *     tests/.
 *
 * All DIN/YouTube arithmetic is float32, the sklearn-port MLP and item2vec are
 * float64, exactly as in the reference.  Compile with -ffp-contract=off.
 */
#ifndef GOCTR_ORACLE_H
#define GOCTR_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ ops -- */
/* model/activation.go:11-16  PRelu32: 0.5*((x-|x|)*slope + (x+|x|)) */
void orc_prelu32(const float* x, float slope, float* out, int n);
/* model/activation.go:57-83  CosineSimilarity over the last axis with size-1 axis broadcast.
 * x is [B,Tx,D], y is [B,Ty,D]; Tx==Ty or one of them is 1.  out is [B,max(Tx,Ty)].
 * returns 0, or -1 for the "shapes not supported" error path. */
int orc_cosine_similarity(const float* x, int Tx, const float* y, int Ty, int B, int D, float* out);
/* model/activation.go:23-50  EucDistance, same broadcasting contract. */
int orc_euc_distance(const float* x, int Tx, const float* y, int Ty, int B, int D, float* out);
/* model/cost.go:9-17, 20-23, 26-29 */
float orc_bce32(const float* y_pred, const float* y_true, int n);
float orc_mse32(const float* y_pred, const float* y_true, int n);
float orc_rms32(const float* y_pred, const float* y_true, int n);

/* ------------------------------------------------------------- roc auc -- */
/* utils/util.go:131-148 -> nn/metrics/ranking.go:13-57,71-118,144-150 */
double orc_roc_auc(const double* score, const double* y, int n);
float orc_roc_auc32(const float* score, const float* y, int n);
/* ROCCurve (ranking.go:71-103): returns number of points written */
int orc_roc_curve(const double* score, const double* y, double pos_label, int n,
                  double* fpr, double* tpr, double* thresholds);

/* ------------------------------------------------------ row assembly ---- */
/* recommend/rcmd.go:462-536 (GetSampleVector) + utils/util.go:22-28 (ConcatSlice32):
 * row = [ userFeat(U) | ub(T*D) | itemEmb(D) | itemFeat(C) ]; ub slot t = emb[ub_ids[t]]
 * for valid ids, zeros for id<0 / id>=V ("missing => zero row"). emb may be NULL
 * (RecSys without ItemEmbedding => all-zero ub/itemEmb, rcmd.go:501). */
void orc_assemble_rows(const float* emb, int64_t V, int D, int T,
                       const int32_t* ub_ids /*[rows,T]*/, const int32_t* item_ids /*[rows]*/,
                       const float* user_feat /*[rows,U]*/, int U,
                       const float* item_feat /*[rows,C]*/, int C,
                       int64_t rows, float* X /*[rows, U+T*D+D+C]*/);

/* ------------------------------------------------------ DIN / YouTube --- */
enum { ORC_DIN = 0, ORC_YOUTUBE = 1 };
enum { ORC_ATT_COSINE = 0, ORC_ATT_EUCLID = 1 };

typedef struct {
  int kind;        /* ORC_DIN | ORC_YOUTUBE */
  int att;         /* ORC_ATT_COSINE (din.go:231-237) | ORC_ATT_EUCLID (din.go:230, commented out) */
  int U, T, D, C;  /* uProfileDim, uBehaviorSize, uBehaviorDim(=iFeatureDim), cFeatureDim */
  int H1, H2;      /* 200, 80 (din.go:17-18, dnn.go consts) */
} orc_ctr_cfg;

/* weights: W0 [I,H1] row-major, W1 [H1,H2], W2 [H2,1], att0 [T] (DIN only); I = U+2D+C */
typedef struct {
  float* W0; float* W1; float* W2; float* att0;
} orc_ctr_weights;

/* gorgonia AdamSolver state (one m,v per learnable), App. B of SURVEY.md */
typedef struct {
  float *m0, *v0, *m1, *v1, *m2, *v2, *ma, *va;
  int iter;
} orc_adam_state;

typedef struct {
  /* float64 like gorgonia's solver fields; cast to float32 per element for float32 tensors, while
   * the bias corrections 1-beta^iter are evaluated in float64 */
  double lr;      /* 0.01   model.go:88 */
  double l2;      /* 1e-4   model.go:88 */
  double beta1, beta2, eps; /* 0.9 0.999 1e-8 (gorgonia defaults) */
  int   adam_div_by_batch;        /* 1: g *= 1/B before the moments (WithBatchSize) */
  int   adam_l2_before_batch_div; /* 1: g += l2*w happens before the 1/B scaling */
} orc_adam_cfg;

/* dropout: mode 0 = off, 1 = explicit masks (m0 [B,H1], m1 [B,H2] of 0/1),
 * 2 = counter-hash mask shared bit-for-bit with the HIP path (seed, step). */
typedef struct {
  int mode; float p0, p1;
  const float* m0; const float* m1;
  uint32_t seed; uint32_t step;
} orc_dropout;

/* the counter hash used for dropout mode 2: returns 1.0f (keep) or 0.0f */
float orc_dropout_keep(uint32_t seed, uint32_t step, uint32_t layer, uint32_t row, uint32_t col, float p);

/* Forward for one batch of B rows taken from dense X (row-major, xcols wide, the four
 * column ranges of rcmd.SampleInfo, rcmd.go:132-137).  Rows >= valid are treated as
 * all-zero rows (model.go:357-371 FillTensorRows).  model/din/din.go:219-323 /
 * model/youtube/dnn.go:162-184.
 * Optional outputs (may be NULL): h0 [B,I], A0 [B,H1] (post-dropout), A1 [B,H2] (post-dropout),
 * gate [B,T] (g), wgt [B,T] (w). */
void orc_ctr_forward(const orc_ctr_cfg* cfg, const orc_ctr_weights* w,
                     const float* X, int xcols, const int ranges[8], int B, int valid,
                     const orc_dropout* drop,
                     float* y_out /*[B]*/, float* h0, float* A0, float* A1, float* gate, float* wgt);

/* One full training step on one batch: forward, BCE (cost.go:9-17), backward (hand-derived,
 * SURVEY App. A.1), no parameter update.  grads laid out like weights.  returns the cost. */
float orc_ctr_loss_grad(const orc_ctr_cfg* cfg, const orc_ctr_weights* w,
                        const float* X, int xcols, const int ranges[8],
                        const float* Y, int B, int valid, const orc_dropout* drop,
                        orc_ctr_weights* grads, float* y_out);

/* gorgonia AdamSolver.Step on all learnables (model.go:88,192). grads are
 * consumed (zeroed like the solver does). */
void orc_ctr_adam_step(const orc_ctr_cfg* cfg, orc_ctr_weights* w, orc_ctr_weights* grads,
                       orc_adam_state* st, const orc_adam_cfg* ac, int batch);

/* model.Train (model.go:27-213): mini-batch loop with zero-padded last batch, Adam after each
 * batch, per-epoch cost = cost of the LAST batch, early stop.  epoch_costs [epochs].
 * dropout mode 2 uses step = global batch counter.  returns epochs actually run. */
int orc_ctr_train(const orc_ctr_cfg* cfg, orc_ctr_weights* w,
                  const float* X, const float* Y, int64_t rows, int xcols, const int ranges[8],
                  int batch, int epochs, int early_stop, const orc_adam_cfg* ac,
                  int drop_mode, float p0, float p1, uint32_t seed,
                  float* epoch_costs);

/* model.Predict (model.go:242-352): batches of `batch` with zero padding, first (end-start)
 * outputs kept.  No dropout (the JSON round trip in dinimpl.go:73-89 drops d0/d1). */
void orc_ctr_predict(const orc_ctr_cfg* cfg, const orc_ctr_weights* w,
                     const float* X, int64_t rows, int xcols, const int ranges[8],
                     int batch, float* y_out);

/* number of host threads the row-parallel loops use (OpenMP); 1 = scalar port */
void orc_set_threads(int n);
int  orc_get_threads(void);

/* -------------------------------------------------- sklearn-port MLP ---- */
/* nn/neural_network/basemlp64.go.  packed theta layout (basemlp64.go:432-463):
 * for each layer i: [ b_i (fanOut) | W_i (fanIn x fanOut row-major) ]. */
enum { ORC_ACT_IDENTITY = 0, ORC_ACT_LOGISTIC = 1, ORC_ACT_TANH = 2, ORC_ACT_RELU = 3 };
enum { ORC_SOLVER_SGD = 0, ORC_SOLVER_ADAM = 1 };

typedef struct {
  int n_layers;            /* len(layerUnits) incl. input and output */
  int units[8];            /* layerUnits */
  int activation;          /* hidden activation */
  double alpha;            /* L2 */
  int batch_normalize;     /* max-abs "batch normalisation" (basemlp64.go:277-308) */
  double weight_decay;     /* theta *= (1-wd) before forward (basemlp64.go:342-346) */
} orc_mlp_cfg;

typedef struct {
  int solver;
  double lr_init, beta1, beta2, eps;      /* adam */
  double momentum; int nesterov;          /* sgd */
  /* state */
  double t; double beta1t, beta2t; double* ms; double* vs; double* velocities; double lr;
} orc_mlp_opt;

size_t orc_mlp_nparams(const orc_mlp_cfg* cfg);
/* forward (basemlp64.go:259-274) on n rows of X [n, units[0]]; out [n, units[last]] logistic. */
void orc_mlp_predict(const orc_mlp_cfg* cfg, const double* theta, const double* X, int n, double* out);
/* backprop (basemlp64.go:340-406): loss + packed grads for one batch of n rows. */
double orc_mlp_loss_grad(const orc_mlp_cfg* cfg, double* theta, const double* X, const double* Y,
                         int n, double* grads);
/* the same on the caller's activation / delta blocks of B rows each (acts[1..L-1], deltas[0..L-2]; acts[0] unused) for a
 * batch of ns <= B rows: ns < B is the reference's SHORT LAST BATCH (quirk Q11, basemlp64.go:790-812) -- rows [ns, B) of
 * acts[1] and deltas[last] are what the previous call left there and take part in the bias / upper-layer gradients. */
double orc_mlp_loss_grad_rows(const orc_mlp_cfg* cfg, double* theta, const double* X, const double* Y, int ns, int B,
                              double* const* acts, double* const* deltas, double* grads);
/* AdamOptimizer64.updateParams (basemlp64.go:1075-1091, per-parameter beta powers) /
 * SGDOptimizer64.updateParams (:1024-1039). */
void orc_mlp_opt_init(orc_mlp_opt* o, int solver, size_t nparams);
void orc_mlp_opt_free(orc_mlp_opt* o);
void orc_mlp_update(orc_mlp_opt* o, double* theta, const double* grads, size_t nparams);
/* fitStochastic (basemlp64.go:729-857) with a GIVEN batch order: X,Y are used in the given row
 * order each epoch (shuffle is the caller's job: perm [max_iter][n] or NULL for identity),
 * a short last batch (n % batch != 0) is trained the reference's way (Q11, stale rows).  loss_curve [max_iter]. tol/no-improve stopping
 * (basemlp64.go:859-895) with constant learning-rate schedule.  returns iterations run. */
int orc_mlp_fit(const orc_mlp_cfg* cfg, double* theta, orc_mlp_opt* opt,
                const double* X, const double* Y, int64_t n, int batch, int max_iter,
                double tol, int n_iter_no_change, const int32_t* perm, double* loss_curve);

/* ------------------------------------------------------------ item2vec -- */
/* feature/embedding/model/modelutil/modelutil.go:21-29 */
typedef struct { uint64_t next; } orc_lcg;
int orc_lcg_next(orc_lcg* g, int value);
/* modelutil.go:32-41 IndexPerThread: out [threads+1] */
void orc_index_per_thread(int threads, int64_t n, int64_t* out);
/* sigmoid_table.go:28-45 */
void orc_sigmoid_table(double* table /*[1000]*/);
double orc_sigmoid_lookup(const double* table, double x);
/* subsample.go:28-43 keep probability */
double orc_subsample_keep(double threshold, int64_t count);

/* corpus/dictionary/huffman.go:23-57 + node/node.go:26-43.
 * counts [V].  Outputs: for each word id its root->leaf path as indices of INNER nodes
 * (inner node k has vector row k in the node matrix, numbered in creation order) and the
 * branch codes; path_off [V+1] CSR offsets into path_nodes/path_codes.  max_depth truncates
 * the path like GetPath (node.go:39-42): the path has at most max_depth nodes INCLUDING the
 * leaf, i.e. at most max_depth-1 (node,code) pairs when not truncated.  returns total pairs. */
int64_t orc_huffman_paths(const int64_t* counts, int64_t V, int max_depth,
                          int64_t* path_off, int32_t* path_nodes, uint8_t* path_codes,
                          int64_t cap);

/* literal O(V^2) restatement of huffman.go:23-57, used to pin the fast builder */
int64_t orc_huffman_paths_slow(const int64_t* counts, int64_t V, int max_depth,
                               int64_t* path_off, int32_t* path_nodes, uint8_t* path_codes,
                               int64_t cap);

typedef struct {
  int dim, window;
  int optimizer;      /* 0 = hierarchical softmax (wordemb.go:13), 1 = negative sampling */
  int model;          /* 0 = skip-gram (wordemb.go:12), 1 = cbow */
  int neg_samples;    /* 5 */
  double init_lr, min_lr;   /* 0.025, 2.5e-6? see options.go:38-58 */
  int64_t update_lr_batch;  /* 100000 */
  int max_depth;      /* 100 */
} orc_w2v_cfg;

/* One deterministic single-stream pass (word2vec.go:198-243 with Goroutines=1):
 * doc [n] word ids, keep_mask [n] (injected subsample trials, NULL = always train),
 * param [V,dim] f64 in/out, node [V-1,dim] f64 in/out (HS inner-node vectors, zero-init) or
 * ctxmat [V,dim] for NS.  lcg is the shared NextRandom state.  lr_state: current lr in/out,
 * trained_cnt in/out (observer counter), corpus_len = unfiltered length (Q17).
 * slice [lo,hi) is the thread's doc slice: window clipping is against the slice (Q18). */
void orc_w2v_train_slice(const orc_w2v_cfg* cfg, const int32_t* doc, int64_t lo, int64_t hi,
                         const uint8_t* keep_mask,
                         double* param, double* node_or_ctx, int64_t V,
                         const int64_t* path_off, const int32_t* path_nodes, const uint8_t* path_codes,
                         const double* sigtab, orc_lcg* lcg,
                         double* lr, int64_t* trained_cnt, int64_t corpus_len);

/* Hogwild run for the CPU baseline: `threads` OpenMP threads over IndexPerThread slices sharing
 * everything unsynchronised exactly like word2vec.go:151-175 (racy by design; only the LCG is kept
 * per-thread-free: it is shared and unsynchronised as in the reference).  */
void orc_w2v_train_hogwild(const orc_w2v_cfg* cfg, const int32_t* doc, int64_t n, int threads,
                           const uint8_t* keep_mask, double* param, double* node_or_ctx, int64_t V,
                           const int64_t* path_off, const int32_t* path_nodes, const uint8_t* path_codes,
                           const double* sigtab, double* lr, int64_t corpus_len);

/* ---- embedding k-NN search (orc_search.c): search.go:92-134, searchutil.go:17-26, embutil.go:21-27 ---- */
double orc_norm64(const double* v, int d);
double orc_cosine64(const double* v1, const double* v2, int d, double n1, double n2);
int orc_knn_search(const double* items, const double* norms, int64_t V, int D, const double* query, double qnorm,
                   int k, int64_t ignore, int64_t* out_idx, double* out_sim, int* out_rank);
/* Q independent searches, OpenMP over the queries (each = orc_knn_search); out_* [Q][k], out_count [Q] */
void orc_knn_search_batch(const double* items, const double* norms, int64_t V, int D, const double* queries, int Q, int k,
                          const int64_t* ignore, int64_t* out_idx, double* out_sim, int* out_rank, int* out_count);

/* ---- user-behaviour cache lookup + per-sample key assembly (orc_ubcache.c): cache.go:71-94, rcmd.go:460-536 ---- */
int64_t orc_ubcache_filter(const int64_t* ts, const int32_t* items, int64_t len, int64_t max_ts, int64_t max_len,
                           int32_t* out);
void orc_assemble_keys(const int64_t* off, const int32_t* seq_items, const int64_t* seq_ts, int64_t n_users,
                       const float* user_table, int U, const float* item_table, int64_t n_items, int C,
                       const int32_t* users, const int32_t* items, const int64_t* ts, int64_t rows, int T,
                       int32_t* ub_ids, float* ufeat, float* cfeat);

#ifdef __cplusplus
}
#endif
/* ---- corpus load / dictionary / IndexedDoc / subsampler table (orc_corpus.c): memory.go:53-102, dictionary.go:70-81,
 * cpsutil.go:58-78, subsample.go:28-43 ---- */
int64_t orc_corpus_build(const int64_t* keys, int64_t n, int64_t min_count, int64_t max_count, int32_t* idoc,
                         int64_t* id2key, int64_t* cfs, int32_t* indexed, int64_t* n_indexed);
void orc_subsample_probs(const int64_t* cfs, int64_t V, double threshold, double* samples);

/* ---- trainable-embedding EXTENSION (orc_embtrain.c): float64 loss and its gradient with respect to the embedding
 * rows; no reference counterpart (embeddings are frozen there), validated by finite differences ---- */
double orc_embtrain_loss_grad(const orc_ctr_cfg* cfg, const orc_ctr_weights* w, const double* E, int64_t V,
                              const int32_t* ub_ids, const int32_t* item_ids, const float* ufeat, const float* cfeat,
                              const float* Y, int B, int valid, const orc_dropout* drop, double* dE /* [V,D] or NULL */);

#endif
