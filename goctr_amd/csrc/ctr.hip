// ctr.hip -- host side of the DIN / YouTube-DNN engine + its C-ABI (include/goctr.h).
//
// Replaces, behind the same operator surface, the gorgonia-executed training / predict loops of
// model/model.go:27-352 for model/din and model/youtube (reference = auxten/go-ctr).  One step =
//   attn_fwd -> 3 x gemm_nn(+epilogue) -> 3 x gemm_nn backward-data -> attn_bwd -> 3 x gemm_tn
//   -> reduce -> [RCCL all-reduce] -> adam
// all on one HIP stream; per-step varying values live in a device-side StepState so the sequence
// can be captured once into a hipGraph and replayed.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <memory>
#include <array>
#include <map>
#include <shared_mutex>

#include "common.h"
#include "ctr_chain.h"
#include "ctr_chain_x3.h"
#include "emb_train.h"
#include "emb_plan.h"
#include "scan.h"
#include "ctr_kernels.h"
#include "ctr_serve.h"
#include "mfma_gemm.h"

using namespace goctr;

struct goctr_emb {
  goctr::Engine* const eng = &goctr::engine();   // the engine (device, streams, arena) the handle was created on
  const uint64_t uid = next_uid();   // what a captured step graph is keyed on (never reused, unlike the host address)
  uint64_t version = 0;              // bumped whenever rows change (goctr_emb_set_rows, embedding training): H0Carry is keyed on it
  int64_t V = 0; int D = 0;
  DevBuf<float> rows;
  // single-call multi-device training (goctr_train_cfg::devices): this table's replicas on engines 1 .. n-1 (owned), and the
  // version of THIS table they were last made equal to
  std::vector<goctr_emb*> reps; uint64_t reps_version = ~0ull;
  // Rows are READ by serving passes on their slots' streams (shared) and WRITTEN on the main stream by goctr_emb_set_rows and
  // by embedding training of any model that was given this table (exclusive).  ev_rows is recorded behind the last queued
  // write: training is asynchronous, a serving pass waits for the event before its launches read the rows.
  std::shared_mutex mu;
  hipEvent_t ev_rows = nullptr; std::atomic<bool> rows_pending{false};
};

struct goctr_dataset {
  goctr::Engine* const eng = &goctr::engine();   // the engine (device, streams, arena) the handle was created on
  const uint64_t uid = next_uid();
  bool id_mode = false;
  int64_t rows = 0;
  bool has_y = false;
  // dense
  DevBuf<float> X; int xcols = 0; int ranges[8] = {0};
  // ids
  DevBuf<int32_t> ub_ids, item_ids; DevBuf<float> ufeat, cfeat; int U = 0, C = 0, T = 0;
  DevBuf<float> Y;
  // single-call multi-device training: shards[r] (on engine r, owned) holds rank r's rows of every global batch of shard_B rows,
  // batch-major, the short last batch zero-padded (model.go:357-371) -- local batch k of rank r = rows [r, r+1) * shard_B / n of
  // global batch k
  std::vector<goctr_dataset*> shards; int shard_B = 0;
  // goctr_train_dense with cfg.devices = n > 1: the caller's HOST rows, valid for the duration of that call only.  Nothing is
  // uploaded to engine 0 (X / Y stay empty): every rank copies ITS rows of every global batch straight from host memory into its
  // shard, on its own device and stream (train_multi) -- round 4 staged all of X on engine 0 and scattered it over xGMI
  const float* host_X = nullptr; const float* host_Y = nullptr;
};

struct StepGraph {
  // One captured step per ping-pong parity of the step state (a step reads slot p and writes slot p^1).
  // b[] only when a communicator splits the step (all-reduce between reduce and Adam).
  hipGraphExec_t a[2] = {nullptr, nullptr}, b[2] = {nullptr, nullptr};
  hipGraphExec_t mid[2] = {nullptr, nullptr};   // data parallel + trainable embeddings: owner side of the sparse exchange + slab reduce
  // ba[p]: b[p] and the NEXT step's a[p ^ 1] as one graph (dense all-reduce only): a step inside a call is then all-reduce +
  // ONE graph launch instead of two -- every boundary between host-issued items costs the GPU ~4 us
  hipGraphExec_t ba[2] = {nullptr, nullptr};
  // multi[p]: multi_steps (even) consecutive steps starting at parity p in ONE graph (single GPU): the boundary between
  // two graph launches costs about two kernel-to-kernel edges; every per-step scalar is device state, so nothing else changes
  // (built together with a[]: a first call in a timed region must not pay for a capture)
  static constexpr int kNMulti = 3;
  int kMulti[kNMulti] = {16, 4, 2};                                        // even, descending (GOCTR_GRAPH_SIZES=a,b,c: experiments)
  hipGraphExec_t multi[kNMulti][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // [size][parity]
  bool multi_on = false;
  // cache key
  // (the captured launches bake in the dataset's / table's device pointers and row count: keyed on the handles'
  // generation ids, not their host addresses -- malloc readily hands a destroyed dataset's address to the next one)
  uint64_t ds = 0, emb = 0; int B = 0; int mode = 0; float p0 = 0, p1 = 0;
  uint32_t seed = 0; double lr = 0, l2 = 0, b1 = 0, b2 = 0, eps = 0; int flags = 0; int world = 1; bool comm = false;
  bool pipelined = false;   // the captured steps are pipelined (StepOpts::pipelined): a replay needs h0 of its first step
  bool fac = false;         // gate_fac_mode() when the steps were captured (their attention launches leave the one factor)
  void destroy() {
    // goctr_train_steps does not synchronise: replays of these execs may still be queued or running, and destroying an
    // exec in flight is not something HIP documents as safe.  The capture that follows a destroy is host-heavy anyway.
    bool any = false;
    for (int k = 0; k < 2; ++k) {
      any = any || a[k] || b[k] || mid[k] || ba[k];
      for (int z = 0; z < kNMulti; ++z) any = any || multi[z][k];
    }
    if (any && engine().inited) (void)hipStreamSynchronize(engine().stream);
    for (int k = 0; k < 2; ++k) {
      if (a[k]) (void)hipGraphExecDestroy(a[k]);
      if (b[k]) (void)hipGraphExecDestroy(b[k]);
      if (mid[k]) (void)hipGraphExecDestroy(mid[k]);
      if (ba[k]) (void)hipGraphExecDestroy(ba[k]);
      for (int z = 0; z < kNMulti; ++z) { if (multi[z][k]) (void)hipGraphExecDestroy(multi[z][k]); multi[z][k] = nullptr; }
      a[k] = b[k] = mid[k] = ba[k] = nullptr;
    }
    multi_on = false;
  }
};

// Where a forward pass keeps its per-row buffers: the training workspace (parity copies of gate / wgt), the model's
// predict workspace, or a serving slot's.  A forward-only launch touches nothing else (the fused chain kernels write
// yhat only; the modular per-layer path also needs P0 / P1).
struct FwdBufs { float* h0; float* gate; float* wgt; float* yhat; float* P0; float* P1; float* fac = nullptr; };
struct FwdWs {
  DevBuf<float> h0, gate, wgt, yhat, P0, P1;
  int B = 0, Ip = 0, T = 0;
  FwdBufs bufs() { return FwdBufs{h0.p, gate.p, wgt.p, yhat.p, P0.p, P1.p}; }
  // (re)allocates for B rows on `st` (zeroed there: h0's pad columns must be 0, never NaN); modular: also P0 / P1
  int ensure(int Bn, int Ipn, int Tn, int H1p, int H2p, bool modular, hipStream_t st) {
    if (Bn <= B && Ipn == Ip && Tn == T && h0.p && (!modular || P0.p)) return 0;
    GOCTR_HIP(hipStreamSynchronize(st));       // launches still reading the old buffers
    B = 0;                                     // (a failure below must not leave the old size next to missing buffers)
    auto z = [&](DevBuf<float>& b, size_t n) -> int {
      if (b.alloc(n, false)) return -1;
      GOCTR_HIP(hipMemsetAsync(b.p, 0, n * sizeof(float), st));
      return 0;
    };
    const size_t Br = (size_t)round_up(Bn, 32);
    if (z(h0, Br * Ipn) || z(gate, Br * Tn) || z(wgt, Br * Tn) || z(yhat, Br)) return -1;
    if (modular && (z(P0, Br * H1p) || z(P1, Br * H2p))) return -1;
    B = Bn; Ip = Ipn; T = Tn;
    return 0;
  }
};

struct goctr_model {
  goctr::Engine* const eng = &goctr::engine();   // the engine (device, streams, arena) the handle was created on
  goctr_ctr_cfg cfg{};
  int I = 0, Ip = 0, H1p = 0, H2p = 0, Dp = 0, Tp = 0;
  int off1 = 0, off2 = 0, offa = 0, nflat = 0;
  DevBuf<float> W, G, Mo, Vo, W1T, W2T, W0sT;
  DevBuf<float> Wimg;   // LDS images of W0 | W1 | W1^T | W0[U:U+D,:]^T (ctr_chain.h), kept in sync by Adam
  // bf16-plane fragment images of the 6-product-split training chain (ctr_chain_x3.h), kept in sync by the Adam kernels
  DevBuf<unsigned short> Wx3; int x3_nch0 = 0;
  CxImages x3_images() {
    CxImages im{nullptr, nullptr, nullptr, nullptr, 0};
    if (!x3_nch0) return im;
    im.nch0 = x3_nch0;
    im.img0 = Wx3.p; im.img1 = im.img0 + cx_img0_elems(x3_nch0); im.img2 = im.img1 + cx_img1_elems(); im.img3 = im.img2 + cx_img2_elems();
    return im;
  }
  float* img(int which) { return Wimg.p + (which == 0 ? 0 : which == 1 ? off1 : which == 2 ? off1 + H1p * H2p : off1 + 2 * H1p * H2p); }
  // per-batch workspace
  int wsB = 0, tnS = 0;
  DevBuf<float> h0, P0, A0, P1, A1, yhat, lossrow, dz2, dz1, dz0, dp, gate, wgt, gfac, slabs0, slabs1, slabs2, attp;
  DevBuf<float> mask0, mask1, slabs3, ones16;
  size_t gw_stride = 0;           // floats between the two parity copies of gate / wgt
  float* gate_p(int par) { return gate.p + (size_t)par * gw_stride; }
  float* gfac_p(int par) { return gfac.p + (size_t)par * gw_stride; }
  float* wgt_p(int par) { return wgt.p + (size_t)par * gw_stride; }
  DevBuf<unsigned int> ra_flag;   // pipelined steps: gstep + 1 of the last step whose att0 update is visible device-wide (reduce_attn_kernel)
  DevBuf<float> yall;          // scores of a whole predict call (one device-to-host copy at the end)
  FwdWs pws;                   // forward-only workspace of goctr_predict_* (the training workspace and its graphs stay untouched)
  DevBuf<StepState> st, pst;   // st: two ping-pong slots, stp = the one the next step reads
  int stp = 0;
  StepState* st_cur() { return st.p + stp; }
  StepState* st_next() { return st.p + (stp ^ 1); }
  DevBuf<float> costs;
  // exclusive: everything that writes weights, optimizer state or the model's own workspaces (training, set_weights,
  // goctr_predict_* on the model's predict workspace); shared: the serving slots' forward passes (ServeSlot below)
  std::shared_mutex mu;
  // recorded on the main stream behind the last queued launch that writes the weights (training is asynchronous): a
  // serving slot's stream waits for it before it reads them
  hipEvent_t ev_weights = nullptr; std::atomic<bool> weights_pending{false};
  StepGraph graph;
  int attp_blocks = 0;
  // trainable-embedding extension (emb_train.h): off unless goctr_model_set_embedding_training(lr > 0)
  float emb_lr = 0.f;
  long long emb_V = 0; int emb_B = 0, emb_world = 0; bool emb_comm = false;
  DevBuf<float> dpv, W0pvT;
  bool w0pv_live = false;         // W0pvT holds the current W0[U:U+2D,:]^T and the Adam kernels keep it current
  // The last launch of a pipelined step computes the NEXT batch's h0 / gates (reduce_attn_kernel); the last step of a
  // goctr_train_steps call computes them for the batch the next call usually starts at.  That call skips its own first attn_fwd
  // (8.5 us + a launch of a call's ~32 us fixed cost) if NOTHING could have touched what those rows were computed from:
  // `gen` counts every entry that locks the model exclusively (weights, state, workspace -- and this model's own calls), the
  // table's version its row updates; dataset and table are identified by their never-reused uids.
  uint64_t gen = 0;
  struct H0Carry { bool valid = false; uint64_t gen = 0, ds_uid = 0, emb_uid = 0, emb_version = 0; int B = 0, stp = 0; long long batch = -1;
                   double beta1 = 0, beta2 = 0; /* (the bias corrections the last loss block left were made with these) */
                   bool fac = false; /* (gate and weight left as one factor: gate_fac_mode) */ } carry;
  bool attn_bwd_in_chain = false;  // launch_chain_x3 -> launch_backward: this step's chain launch wrote the att0 terms
  bool dpv_from_chain = false;    // the step's chain launch wrote dpv itself (launch_chain_x3): no dpv GEMM in this step
  // round 6: the step's chain launch left dW2 / the att0 terms as per-tile sums (tile_dw2 / tile_att0; ctr_chain_x3.h): the
  // weight-gradient launch only adds the tiles up (mfma_gemm.h tn_tile_sum_body) and A1, dz2, attp are not written at all
  bool dw2_from_chain = false, att0_from_chain = false;
  bool att0_early = false;       // this step's weight-gradient launch has already updated att0 (ctr_chain_x3.h att0_early_body): launch_backward -> launch_reduce_part
  DevBuf<float> tile_dw2, tile_att0;
  DevBuf<unsigned int> emb_mark, emb_rank, emb_tiles;
  DevBuf<unsigned long long> emb_total;
  DevBuf<long long> emb_accum;
  DevBuf<int> emb_slot_id;
  long long emb_Vw = 0;           // rows of one owner's bucket in the (owner-major) mark / rank index space
  // per-batch sparse plan of the id-major update (emb_train.h, "Round 3"): built once per (dataset, batch, vocabulary, world)
  struct EmbPlan {
    bool valid = false; uint64_t ds = 0; long long V = 0; int B = 0, W = 0, T = 0;
    DevBuf<int> pair, pslot, pid, slot_id; DevBuf<unsigned int> slot_off; DevBuf<long long> pair_off, slot_base;
    long long nb = 0, max_pairs = 0, max_slots = 0, total_pairs = 0, total_slots = 0;
    double build_ms = 0;         // host wall time of the build (goctr_model_emb_plan_build_ms)
    EmbPlanView view() const { return EmbPlanView{pair.p, pslot.p, pid.p, pair_off.p, slot_id.p, slot_off.p, slot_base.p}; }
  } plan;
  DevBuf<float> emb_dx, emb_gsum;  // emb_coef's per-pair row gradients [B, T, D] and item-row gradients [B, D]
  // fixed-size exchange (emb_train.h, end): exact bounds from the plan, no host read-back between the collectives
  bool ex_fixed = false; int ex_S = 0, ex_R = 0;
  DevBuf<int> ex_bucket_off, ex_send_ids, ex_recv_ids; DevBuf<long long> ex_send_rows, ex_recv_rows;
  ReduceArgs pend_ra{};            // launch_backward(stage 1) -> (stage 2)
  bool pend_no_costs = false;      // goctr_train_steps: the caller does not read this call's costs
  bool pend_retarget = false; long long pend_batch_idx = 0, pend_n_batches = 1;   // goctr_train_steps -> run_steps' state-preparation launch
  // bucketed exchange (data parallel): bucket bounds / counts, received pairs, the owner's reduction, the gathered deltas
  DevBuf<int> ex_off, ex_cnt, ex_allcnt, ex_rids, ex_red_ids, ex_nred, ex_allnred, ex_gids;
  DevBuf<long long> ex_rrows, ex_red;
  DevBuf<float> ex_delta, ex_gdelta;
  DevBuf<unsigned long long> ex_red_total;
  double ex_bytes_last = 0;       // bytes this rank SENT in the last step's exchange
  // single-call multi-device training: replicas on engines 1 .. n-1 (owned; reps[0] unused) and this model's `gen` after the
  // last call that left them bit-identical to it (anything else that locked the model since then forces a re-broadcast)
  std::vector<goctr_model*> reps; uint64_t reps_gen = ~0ull;
};

namespace {

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// Slab heights of the weight-gradient launch.  fp32 MFMA work of co-resident workgroups serialises on a SIMD,
// so the launch is sized to ONE equally expensive workgroup per CU: the 3-tile problems (dW0: ceil(Ip/48)
// blocks, dW1: ceil(H2p/48) blocks per slab) take `rows` batch rows per workgroup, the one-tile problems
// (dW2, datt0: a third of the MFMAs per row) take 3x as many.
#ifndef GOCTR_TN_CH
#define GOCTR_TN_CH 32
#endif
struct TnSchedule { int rows, S, rows_light, S_light; };
TnSchedule tn_schedule(const goctr_model* m, int B) {
  TnSchedule t{};
  const int heavy = (int)cdiv(m->Ip / 16, 3) + (int)cdiv(m->H2p / 16, 3);
  const int light = m->cfg.kind == GOCTR_DIN ? 2 : 1;
  int cus = engine().compute_units > 0 ? engine().compute_units : 256;
  const int lf = 25;   // light slab height = lf/10 x the heavy one
  // workgroups(rows) = heavy*ceil(B/rows) + light*ceil(B/(lf rows)) <= cus ; smallest such rows (multiple of 4)
  int rows = 32;
  for (;; rows += 4) {
    const long wgs = (long)heavy * cdiv(B, rows) + (long)light * cdiv(B, round_up(rows * lf / 10, 4));
    if (wgs <= cus || rows >= B) break;
  }
  t.rows = rows; t.S = (int)cdiv(B, rows);
  t.rows_light = round_up(rows * lf / 10, 4); t.S_light = (int)cdiv(B, t.rows_light);
  return t;
}
// The wide-block weight-gradient kernel (mfma_gemm.h, gemm_tn_multi_x3w_kernel): shapes it covers and its slab heights.
// Workgroups of the two heavy problems stage different numbers of operand columns per chunk (the stagers bound the chunk
// time), so each problem gets its own slab height, in whole 32-row chunks: the pair (c0, c1) that minimises the longest
// workgroup subject to one workgroup per CU.
struct TnWide { bool ok; int ktw0, kblocks0, nbt; int rows0, S0, rows1, S1, rowsL, SL; };
TnWide tn_schedule_wide_search(const goctr_model* m, int B, int nsum);
// the search is O((B/32)^2) (65 k iterations at B = 8192): graph replay hides it, the eager paths (data-parallel embedding
// training, profiling, GOCTR_NO_GRAPH) would pay it on every step -- cached per shape and experiment-knob setting
// slabs of a "sum problem" of the weight-gradient launch (the chain launch left per-tile sums: tn_tile_sum_body): 8, one per
// thread of the reduce launch's 8-thread groups
constexpr int TN_SUM_SLABS = 8;
// nsum: how many of the light problems (dW2, att0) are sums over the chain launch's per-tile results this step
TnWide tn_schedule_wide(const goctr_model* m, int B, int nsum = 0) {
  static std::mutex mu;
  static std::map<std::array<int, 7>, TnWide> cache;
  const std::array<int, 7> key{B, m->Ip, m->H1p, m->H2p, m->cfg.kind, engine().compute_units, nsum};
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const TnWide w = tn_schedule_wide_search(m, B, nsum);
  cache.emplace(key, w);
  return w;
}
TnWide tn_schedule_wide_search(const goctr_model* m, int B, int nsum) {
  TnWide w{};
  const int kt0 = m->Ip / 16, nt = m->H1p / 16, kt1 = m->H2p / 16;
  w.ok = (kt0 == 9 || kt0 == 15) && kt1 == 5 && nt > 8 && nt <= 16;
  if (!w.ok) return w;
  w.ktw0 = kt0 == 9 ? 9 : 8; w.kblocks0 = (int)cdiv(kt0, w.ktw0); w.nbt = (int)cdiv(nt, 2);
  const int cols0 = w.ktw0 * 16 + w.nbt * 16, cols1 = kt1 * 16 + w.nbt * 16;
  const int colsL = 16 + kt1 * 16;
  const int blocks0 = w.kblocks0 * 2, blocks1 = 2, light = m->cfg.kind == GOCTR_DIN ? 2 : 1;
  const int cus = engine().compute_units > 0 ? engine().compute_units : 256;
  // cost of a workgroup in "staged columns": chunks x columns per chunk + a fixed part (start, first chunk, slab stores)
  const int fix0 = 512, fix1 = 384, fixL = 384;
  long best = -1; int bc0 = 0, bc1 = 0, brl = 0;
  const int cmax = (int)cdiv(B, 32);
  for (int c0 = 1; c0 <= cmax; ++c0)
    for (int c1 = 1; c1 <= cmax; ++c1) {
      long t = std::max((long)c0 * cols0 + fix0, (long)c1 * cols1 + fix1);
      if (best >= 0 && t >= best) continue;
      // the one-tile problems take what is left of the chip; their slab height follows
      const long heavy = (long)blocks0 * cdiv(B, c0 * 32) + (long)blocks1 * cdiv(B, c1 * 32);
      // (nsum of the light problems are sums over the chain launch's per-tile results: TN_SUM_SLABS small workgroups each)
      const int mlight = light - nsum;
      const long room = cus - heavy - (long)nsum * TN_SUM_SLABS;
      const long left = mlight > 0 ? room / mlight : 1;
      if (room < 0 || left < 1) continue;
      const int rl = std::max(32, round_up((int)cdiv(B, left), 4));     // (the slab buffers hold ceil(B / 32) slabs)
      if (mlight > 0) t = std::max(t, (long)cdiv(rl, 32) * colsL + fixL);   // (a sum workgroup is a few hundred loads: never the longest)
      if (best >= 0 && t >= best) continue;
      best = t; bc0 = c0; bc1 = c1; brl = rl;
    }
  if (best < 0) { bc0 = bc1 = cmax; brl = B; }
  w.rows0 = bc0 * 32; w.S0 = (int)cdiv(B, w.rows0);
  w.rows1 = bc1 * 32; w.S1 = (int)cdiv(B, w.rows1);
  w.rowsL = brl; w.SL = (int)cdiv(B, w.rowsL);
  return w;
}
int tn_max_slabs(int B) { return (int)cdiv(B, 32); }
// does launch_backward take the wide bf16-split weight-gradient launch for this model and batch?  (launch_chain_x3 asks: only
// that launch knows how to add up per-tile sums)
bool dw_wide_path(const goctr_model* m, int B) {
  int nt_max = m->H1p / 16;
  if (m->H2p / 16 > nt_max) nt_max = m->H2p / 16;
  if (m->cfg.kind == GOCTR_DIN && m->Tp / 16 > nt_max) nt_max = m->Tp / 16;
  const bool multi = nt_max <= 16 && gemm_tn_multi_fits<3, GOCTR_TN_CH>(nt_max);
  return multi && GOCTR_TN_CH == 32 && tn_schedule_wide(m, B).ok;
}

int ensure_workspace(goctr_model* m, int B) {
  if (m->wsB >= B && m->tnS > 0) return 0;
  const int S = tn_max_slabs(B);   // upper bound over every schedule tn_schedule() can pick
  m->tnS = S;
  if (m->h0.alloc((size_t)B * m->Ip)) return -1;
  if (m->P0.alloc((size_t)B * m->H1p)) return -1;
  if (m->A0.alloc((size_t)B * m->H1p)) return -1;
  if (m->P1.alloc((size_t)B * m->H2p)) return -1;
  if (m->A1.alloc((size_t)B * m->H2p)) return -1;
  if (m->yhat.alloc((size_t)B)) return -1;
  if (m->lossrow.alloc((size_t)B)) return -1;
  if (m->dz2.alloc((size_t)B * 16)) return -1;
  if (m->dz1.alloc((size_t)B * m->H2p)) return -1;
  if (m->dz0.alloc((size_t)B * m->H1p)) return -1;
  if (m->dp.alloc((size_t)B * m->Dp)) return -1;
  // two copies, by the parity of the step state: in pipelined graphs the next step's attn_fwd writes its gates while this
  // step's backward still reads its own
  if (m->gate.alloc((size_t)2 * B * m->cfg.T)) return -1;
  if (m->wgt.alloc((size_t)2 * B * m->cfg.T)) return -1;
  if (m->gfac.alloc((size_t)2 * B * m->cfg.T)) return -1;
  m->gw_stride = (size_t)B * m->cfg.T;
  if (m->ra_flag.alloc(1)) return -1;
  if (m->slabs0.alloc((size_t)S * m->Ip * m->H1p)) return -1;
  if (m->slabs1.alloc((size_t)S * m->H1p * m->H2p)) return -1;
  if (m->slabs2.alloc((size_t)S * m->H2p * 16)) return -1;
  m->attp_blocks = (int)cdiv(B, ATTN_BWD_WAVES);
  if (m->attp.alloc((size_t)B * m->Tp)) return -1;               // dgs [B, Tp]
  if (m->slabs3.alloc((size_t)S * 16 * m->Tp)) return -1;        // att0 gradient slabs (row 0 of 16)
  if (m->tile_dw2.alloc((size_t)S * m->H2p) || m->tile_att0.alloc((size_t)S * m->Tp)) return -1;   // (S = ceil(B / 32) tiles)
  {
    std::vector<float> ones((size_t)B * 16, 0.f);
    for (int r = 0; r < B; ++r) ones[(size_t)r * 16] = 1.0f;
    if (m->ones16.alloc(ones.size(), false) || m->ones16.upload(ones.data(), ones.size())) return -1;
  }
  m->wsB = B;
  m->graph.destroy();
  return 0;
}

RowSource make_source(const goctr_dataset* d, const goctr_emb* e) {
  RowSource s{};
  s.rows = d->rows;
  s.Y = d->has_y ? d->Y.p : nullptr;
  s.id_mode = d->id_mode ? 1 : 0;
  if (d->id_mode) {
    s.emb = e->rows.p; s.V = e->V;
    s.ub_ids = d->ub_ids.p; s.item_ids = d->item_ids.p; s.ufeat = d->ufeat.p; s.cfeat = d->cfeat.p;
  } else {
    s.X = d->X.p; s.xcols = d->xcols;
    s.r_u = d->ranges[0]; s.r_ub = d->ranges[2]; s.r_v = d->ranges[4]; s.r_c = d->ranges[6];
  }
  return s;
}

// dynamic LDS above 64 KiB needs an explicit opt-in per kernel
template <class K>
int allow_big_lds(K kernel) {
  GOCTR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(160 * 1024)));
  return 0;
}

int serve16_attributes() {
#define GOCTR_S16(L, H) (allow_big_lds(ctr_serve16_kernel<L, 1, H>) || allow_big_lds(ctr_serve16_kernel<L, 2, H>) || allow_big_lds(ctr_serve16_kernel<L, 3, H>))
  return (GOCTR_S16(2, 10) || GOCTR_S16(2, 15) || GOCTR_S16(4, 10) || GOCTR_S16(4, 15) || GOCTR_S16(16, 10) || GOCTR_S16(16, 15)) ? -1 : 0;
#undef GOCTR_S16
}

template <class Epi>
int launch_nn(int kid, const float* A, int lda, const float* Bm, int ldb, int M, int Kp, int Np, Epi epi) {
  const int NT = Np / 16;
  // wave grid: prefer >= 1 workgroup per CU (M/32 rows each) when the columns can be split
  int WN = NT >= 2 ? 2 : 1;
  while (WN < 4 && (int)cdiv(NT, WN) > GEMM_NN_NTW) WN *= 2;
  int ntw = (int)cdiv(NT, WN);
  if (ntw > GEMM_NN_NTW) ntw = GEMM_NN_NTW;   // more column blocks in grid.y
  ntw = ntw <= 1 ? 1 : (ntw <= 3 ? 3 : (ntw <= 4 ? 4 : 7));  // instantiated tile counts
  const int WM = 4 / WN;
  dim3 grid((unsigned)cdiv(M, 16 * WM), (unsigned)cdiv(NT, WN * ntw));
  const int ncols_blk = std::min(NT, WN * ntw) * 16;
  (void)ncols_blk;
  const int ncols_alloc = WN * ntw * 16;
  const int KPH = gemm_nn_phase_rows<float>(Kp, ncols_alloc);
  const size_t lds = gemm_nn_lds_bytes<float>(KPH, ncols_alloc);
  hipStream_t st = engine().active;
  ProfScope ps(kid);
  // B fits one LDS phase and there are many more row tiles than CUs: the persistent variant parks B once per workgroup
  if (KPH >= Kp && ntw == 4 && (int)grid.x >= 2 * engine().compute_units) {
    const dim3 g2((unsigned)engine().compute_units, grid.y);
    hipLaunchKernelGGL((gemm_nn_rows_kernel<float, Epi, 4>), g2, dim3(256), lds, st, A, lda, Bm, ldb, M, Kp, Np, WN, epi);
    GOCTR_HIP(hipGetLastError());
    return 0;
  }
#define GOCTR_NN(N) hipLaunchKernelGGL((gemm_nn_kernel<float, Epi, N>), grid, dim3(256), lds, st, A, lda, Bm, ldb, M, Kp, Np, WN, KPH, epi)
  switch (ntw) {
    case 1: GOCTR_NN(1); break;
    case 3: GOCTR_NN(3); break;
    case 4: GOCTR_NN(4); break;
    default: GOCTR_NN(7); break;
  }
#undef GOCTR_NN
  GOCTR_HIP(hipGetLastError());
  return 0;
}

template <int KTW, int NTW, int CH>
int launch_tn_cfg(const float* A, int lda, int KT, const float* Dm, int ldd, int NT, int M, int rows_per_wg,
                  int WK, int WN, float* slabs, size_t slab_stride) {
  const int S = (int)cdiv(M, rows_per_wg);
  const int nthreads = 64 * WK * WN;
  // staging registers must cover one chunk
  GOCTR_CHECK((size_t)CH * (WK * KTW * 16 / 4) <= (size_t)GEMM_TN_MAXVA * nthreads &&
              (size_t)CH * (WN * NTW * 16 / 4) <= (size_t)GEMM_TN_MAXVD * nthreads,
              "gemm_tn: chunk does not fit the staging registers (WK=%d WN=%d)", WK, WN);
  dim3 grid(S, (unsigned)cdiv(KT, WK * KTW), (unsigned)cdiv(NT, WN * NTW));
  hipLaunchKernelGGL((gemm_tn_kernel<float, KTW, NTW, CH>), grid, dim3(nthreads),
                     gemm_tn_lds_bytes<float>(WK * KTW, WN * NTW, CH), engine().active, A, lda, KT, Dm, ldd, NT, M,
                     rows_per_wg, WK, WN, slabs, slab_stride);
  GOCTR_HIP(hipGetLastError());
  return 0;
}

int launch_tn(int kid, const float* A, int lda, int KT, const float* Dm, int ldd, int NT, int M, int rows_per_wg,
              float* slabs, size_t slab_stride) {
  ProfScope ps(kid);
  if (NT <= 3) {
    // narrow D (dz2: 1 tile): k-tiles across up to 4 waves
    const int WK = std::min(4, (int)cdiv(KT, 4));
    return launch_tn_cfg<4, 3, 16>(A, lda, KT, Dm, ldd, NT, M, rows_per_wg, WK, 1, slabs, slab_stride);
  }
  if (NT <= 6) {
    // dW1-like (13 x 5 tiles): 4 x 2 waves of 4 x 3 tiles
    const int WK = std::min(4, (int)cdiv(KT, 4)), WN = std::min(2, (int)cdiv(NT, 3));
    return launch_tn_cfg<4, 3, 32>(A, lda, KT, Dm, ldd, NT, M, rows_per_wg, WK, WN, slabs, slab_stride);
  }
  // dW0-like (9 x 13 tiles): k-blocks of 3 tiles in grid.y, 4 waves of 3 x 4 tiles across N
  const int WN = std::min(4, (int)cdiv(NT, 4));
  return launch_tn_cfg<3, 4, 32>(A, lda, KT, Dm, ldd, NT, M, rows_per_wg, 1, WN, slabs, slab_stride);
}

// opt every GEMM instantiation into > 64 KiB of dynamic LDS up front (never inside a stream capture)
int init_kernel_attrs() {
  bool& done = engine().kernel_attrs_done;     // (function attributes are per device)
  if (done) return 0;
#define GOCTR_NN_ATTR(E) (allow_big_lds(gemm_nn_kernel<float, E, 1>) || allow_big_lds(gemm_nn_kernel<float, E, 3>) || \
                          allow_big_lds(gemm_nn_kernel<float, E, 4>) || allow_big_lds(gemm_nn_kernel<float, E, 7>) || \
                          allow_big_lds(gemm_nn_rows_kernel<float, E, 4>))
  if (GOCTR_NN_ATTR(EpiSigDrop) || GOCTR_NN_ATTR(EpiOut) || GOCTR_NN_ATTR(EpiDSig) || GOCTR_NN_ATTR(EpiStore) ||
      allow_big_lds(ctr_chain_kernel<7, 5, 0>) || allow_big_lds(ctr_chain_kernel<7, 5, 1>) ||
      allow_big_lds(ctr_chain_kernel<7, 5, 2>) || allow_big_lds(ctr_fwd16_kernel<4, 5>) || allow_big_lds(emb_grad_kernel<16, 0, true>) || allow_big_lds(emb_grad_kernel<16, 1, true>) || allow_big_lds(emb_grad_kernel<16, 2, true>) || allow_big_lds(emb_grad_kernel<32, 0, true>) ||
      allow_big_lds(emb_grad_kernel<32, 1, true>) || allow_big_lds(emb_grad_kernel<32, 2, true>) || allow_big_lds(emb_grad_kernel<64, 0, true>) || allow_big_lds(emb_grad_kernel<64, 1, true>) ||
      allow_big_lds(emb_grad_kernel<64, 2, true>) || allow_big_lds(gemm_tn_kernel<float, 4, 3, 16>) || allow_big_lds(gemm_tn_kernel<float, 4, 3, 32>) ||
      allow_big_lds(gemm_tn_kernel<float, 3, 4, 32>) || allow_big_lds(gemm_tn_multi_kernel<3, 4, GOCTR_TN_CH>) ||
      allow_big_lds(gemm_tn_multi_x3_kernel<3, 4>) || allow_big_lds(gemm_tn_multi_x3w_kernel<9, 5>) || allow_big_lds(gemm_tn_multi_x3w_kernel<8, 5>) ||
      allow_big_lds(gemm_tn_multi_x3w_att0_kernel<9, 5>) || allow_big_lds(gemm_tn_multi_x3w_att0_kernel<8, 5>) || allow_big_lds(ctr_chain_x3_kernel<2>) || allow_big_lds(ctr_chain_x3_kernel<9>) ||
      allow_big_lds(ctr_chain_x3_kernel<15>) || chain_x3_fwd_attributes() || fwd4_attributes() ||
      serve16_attributes()) return -1;
  done = true;
  return 0;
}

int launch_attn_fwd(const AttnArgs& a) {
  ProfScope ps(GOCTR_K_ATTN_FWD);
  dim3 grid((unsigned)cdiv(a.B, 4)), blk(256);
  hipStream_t st = engine().active;
  const bool vec4 = a.src.id_mode && a.D % 4 == 0;
  const int groups = vec4 ? a.D / 4 : a.D;  // lanes needed per row
  // compile-time mode for the shapes that matter (id mode, every lane owns 4 in-range columns)
  // (the compile-time modes address table rows with 32-bit byte offsets: tables below 4 GB)
  const bool small_table = (unsigned long long)(a.src.V + 1) * (unsigned long long)a.D * 4ull < (1ull << 32);
  const int fast = !(vec4 && small_table && groups * 4 == a.D && (groups & (groups - 1)) == 0) ? 0
                   : a.kind != GOCTR_DIN ? 1 : (a.att == GOCTR_ATT_COSINE ? 2 : 3);
  if (ps.on) {   // the symbol the dispatch below selects (goctr_prof_kernel)
    static char sym[48];
    int L = 1;
    while (L < groups) L *= 2;
    if (!vec4 && L < 8) L = 8;
    if (a.src.k_users) snprintf(sym, sizeof sym, "attn_fwd_keys_kernel<%d,%d>", std::min(L, 64), fast);
    else snprintf(sym, sizeof sym, "attn_fwd_kernel<%d,%d,%d>", vec4 ? 4 : 1, std::min(L, 64), (fast && groups <= 16) ? fast : 0);
    prof_note_kernel(GOCTR_K_ATTN_FWD, sym);
  }
  if (a.src.k_users) {
    // serving pass in key mode (serve_keys_pass): only the compile-time shapes have a key variant -- the caller checked
    if (!(fast && groups <= 16)) { set_error("attn_fwd: key mode needs an id-mode fast shape"); return -1; }
#define GOCTR_ATTN_KEYS(L)                                                                         \
  do {                                                                                             \
    if (fast == 1) hipLaunchKernelGGL((attn_fwd_keys_kernel<L, 1>), grid, blk, 0, st, a);          \
    else if (fast == 2) hipLaunchKernelGGL((attn_fwd_keys_kernel<L, 2>), grid, blk, 0, st, a);     \
    else hipLaunchKernelGGL((attn_fwd_keys_kernel<L, 3>), grid, blk, 0, st, a);                    \
  } while (0)
    if (groups == 1) GOCTR_ATTN_KEYS(1);
    else if (groups == 2) GOCTR_ATTN_KEYS(2);
    else if (groups == 4) GOCTR_ATTN_KEYS(4);
    else if (groups == 8) GOCTR_ATTN_KEYS(8);
    else GOCTR_ATTN_KEYS(16);
#undef GOCTR_ATTN_KEYS
    GOCTR_HIP(hipGetLastError());
    return 0;
  }
#define GOCTR_ATTN_FWD(V, L) hipLaunchKernelGGL((attn_fwd_kernel<V, L, 0>), grid, blk, 0, st, a)
#define GOCTR_ATTN_FWD_FAST(L)                                                                     \
  do {                                                                                             \
    if (fast == 1) hipLaunchKernelGGL((attn_fwd_kernel<4, L, 1>), grid, blk, 0, st, a);            \
    else if (fast == 2) hipLaunchKernelGGL((attn_fwd_kernel<4, L, 2>), grid, blk, 0, st, a);       \
    else hipLaunchKernelGGL((attn_fwd_kernel<4, L, 3>), grid, blk, 0, st, a);                      \
  } while (0)
  if (fast && groups <= 16) {
    if (groups == 1) GOCTR_ATTN_FWD_FAST(1);
    else if (groups == 2) GOCTR_ATTN_FWD_FAST(2);
    else if (groups == 4) GOCTR_ATTN_FWD_FAST(4);
    else if (groups == 8) GOCTR_ATTN_FWD_FAST(8);
    else GOCTR_ATTN_FWD_FAST(16);
  } else
  if (vec4) {
    if (groups <= 1) GOCTR_ATTN_FWD(4, 1);
    else if (groups <= 2) GOCTR_ATTN_FWD(4, 2);
    else if (groups <= 4) GOCTR_ATTN_FWD(4, 4);
    else if (groups <= 8) GOCTR_ATTN_FWD(4, 8);
    else if (groups <= 16) GOCTR_ATTN_FWD(4, 16);
    else if (groups <= 32) GOCTR_ATTN_FWD(4, 32);
    else GOCTR_ATTN_FWD(4, 64);
  } else {
    if (groups <= 8) GOCTR_ATTN_FWD(1, 8);
    else if (groups <= 16) GOCTR_ATTN_FWD(1, 16);
    else if (groups <= 32) GOCTR_ATTN_FWD(1, 32);
    else GOCTR_ATTN_FWD(1, 64);
  }
#undef GOCTR_ATTN_FWD_FAST
#undef GOCTR_ATTN_FWD
  GOCTR_HIP(hipGetLastError());
  return 0;
}

int launch_attn_bwd(const AttnBwdArgs& a, int blocks) {
  ProfScope ps(GOCTR_K_ATTN_BWD);
  dim3 grid((unsigned)blocks), blk(64 * ATTN_BWD_WAVES);
  hipStream_t st = engine().active;
  const size_t lds = 0;
  const bool vec4 = a.src.id_mode && a.D % 4 == 0;
  const int groups = vec4 ? a.D / 4 : a.D;
  const bool fast = vec4 && groups * 4 == a.D && (groups & (groups - 1)) == 0 && groups <= 16;
#define GOCTR_ATTN_BWD(V, L) hipLaunchKernelGGL((attn_bwd_kernel<V, L, 0>), grid, blk, lds, st, a)
#define GOCTR_ATTN_BWD_FAST(L) hipLaunchKernelGGL((attn_bwd_kernel<4, L, 1>), grid, blk, lds, st, a)
  if (fast) {
    if (groups == 1) GOCTR_ATTN_BWD_FAST(1);
    else if (groups == 2) GOCTR_ATTN_BWD_FAST(2);
    else if (groups == 4) GOCTR_ATTN_BWD_FAST(4);
    else if (groups == 8) GOCTR_ATTN_BWD_FAST(8);
    else GOCTR_ATTN_BWD_FAST(16);
  } else
  if (vec4) {
    if (groups <= 1) GOCTR_ATTN_BWD(4, 1);
    else if (groups <= 2) GOCTR_ATTN_BWD(4, 2);
    else if (groups <= 4) GOCTR_ATTN_BWD(4, 4);
    else if (groups <= 8) GOCTR_ATTN_BWD(4, 8);
    else if (groups <= 16) GOCTR_ATTN_BWD(4, 16);
    else if (groups <= 32) GOCTR_ATTN_BWD(4, 32);
    else GOCTR_ATTN_BWD(4, 64);
  } else {
    if (groups <= 8) GOCTR_ATTN_BWD(1, 8);
    else if (groups <= 16) GOCTR_ATTN_BWD(1, 16);
    else if (groups <= 32) GOCTR_ATTN_BWD(1, 32);
    else GOCTR_ATTN_BWD(1, 64);
  }
#undef GOCTR_ATTN_BWD_FAST
#undef GOCTR_ATTN_BWD
  GOCTR_HIP(hipGetLastError());
  return 0;
}

struct StepOpts {
  bool train = true;        // false: forward only (predict)
  bool update = true;       // false: stop after the reduce (parity entry)
  int drop_mode = 0; float p0 = 0, p1 = 0; uint32_t seed = 0;
  const goctr_train_cfg* tc = nullptr;
  // pipelined steps (graph replay, single GPU): a step's h0 was computed by the PREVIOUS step's last launch
  // (reduce_attn_kernel, ctr_kernels.h) -- launch_forward skips attn_fwd, launch_backward ends with the merged launch
  bool pipelined = false;
};

// the fused chain kernel covers the reference's fixed hidden widths (200 -> 13 tiles, 80 -> 5 tiles)
bool chain_ok(const goctr_model* m) {
  const int nt0 = m->H1p / 16;
  return (nt0 == 13 || nt0 == 14) && m->H2p == 80 && (m->cfg.kind != GOCTR_DIN || m->Dp <= 16 * CHAIN_NDP) &&
         m->Ip <= 16 * CHAIN_HV &&
         chain_lds_bytes<5>(m->Ip, m->H1p, m->H2p) <= 160u * 1024u &&
         env_int("GOCTR_NO_CHAIN", 0) == 0;
}

// the bf16-split training chain (ctr_chain_x3.h) covers the reference's hidden widths with Ip in {144, 240} (cfg3 DIN /
// the MovieLens-100k defaults, cfg4 YouTube) and the small test shape Ip = 32; hash dropout or none
bool chain_x3_shape_ok(const goctr_model* m) {
  const int nch0 = m->Ip / 16;
  return m->H1p == 208 && m->H2p == 80 && (nch0 == 2 || nch0 == 9 || nch0 == 15) &&
         (m->cfg.kind != GOCTR_DIN || m->Dp <= 32);
}
// training steps, and predict launches large enough to give every CU a 32-row tile (the forward-only variant; smaller
// predict launches are latency-bound and keep ctr_fwd16_kernel)
bool chain_x3_ok(const goctr_model* m, const StepOpts& o, int B) {
  if (m->x3_nch0 == 0) return false;
  if (o.train) return o.drop_mode != 1;
  const int cus = engine().compute_units > 0 ? engine().compute_units : 256;
  return cdiv(B, 32) >= cus;
}

int rebuild_x3_images(goctr_model* m) {
  if (!m->x3_nch0) return 0;
  hipLaunchKernelGGL(x3_build_images_kernel, dim3((unsigned)cdiv(m->off2, 256)), dim3(256), 0, engine().stream, m->W.p, m->off1, m->off2,
                     m->H1p, m->H2p, m->cfg.U, m->cfg.D, m->x3_images());
  GOCTR_HIP(hipGetLastError());
  return 0;
}

template <int NCH0>
void launch_chain_x3_n(const ChainX3Args& a, dim3 grid, hipStream_t s, bool fwd) {
  if (fwd) launch_chain_x3_fwd(NCH0, a, grid, s);       // (ctr_fwd.hip: its own translation unit, see there)
  else hipLaunchKernelGGL((ctr_chain_x3_kernel<NCH0, false>), grid, dim3(512), chain_x3_lds_bytes<NCH0>(), s, a);
}

bool emb_plan_active(const goctr_model* m) { return m->emb_lr > 0.f && m->plan.valid; }

bool gate_fac_mode(const goctr_model* m, const RowSource& src, const StepOpts& o, int B);
int launch_chain_x3(goctr_model* m, const RowSource& src, int B, const StepOpts& o, const StepState* st, const FwdBufs& fb) {
  const goctr_ctr_cfg& c = m->cfg;
  Engine& e = engine();
  const uint32_t row_off = (uint32_t)(e.eff_rank() * B);
  const bool drop = o.drop_mode == 2;
  const CxImages im = m->x3_images();
  ChainX3Args a{};
  a.h0 = fb.h0; a.Ip = m->Ip;
  a.img0 = im.img0; a.img1 = im.img1; a.img2 = im.img2; a.img3 = im.img3; a.w2 = m->W2T.p;
  a.H1 = c.H1; a.H2 = c.H2; a.H1p = m->H1p; a.H2p = m->H2p; a.Dp = m->Dp; a.B = B; a.kind = c.kind;
  a.d0 = DropCfg{drop && o.p0 > 0 ? 2 : 0, o.p0, nullptr, c.H1, o.seed, 0u, row_off};
  a.d1 = DropCfg{drop && o.p1 > 0 ? 2 : 0, o.p1, nullptr, c.H2, o.seed, 1u, row_off};
  a.st = st; a.Y = src.Y; a.rows = src.rows; a.inv_bglobal = 1.0f / (float)(B * e.eff_world());
  a.A0 = m->A0.p; a.A1 = m->A1.p; a.dz0 = m->dz0.p; a.dz1 = m->dz1.p; a.dz2 = m->dz2.p; a.dp = m->dp.p;   // (forward only: none of these is touched)
  // trainable embeddings, DIN, 2 D <= 32: the 32-wide dp product of this kernel also yields d cost / d candidate-item segment
  // (IMG3 holds W0[U : U+2D]^T) -- it writes dpv = [dp | dvh] itself and the step needs no GEMM launch for it (6.9 us at cfg3)
  m->dpv_from_chain = o.train && emb_plan_active(m) && src.id_mode && c.kind == GOCTR_DIN && 2 * c.D <= 32;
  if (m->dpv_from_chain) { a.dp = m->dpv.p; a.Dp = round_up(2 * c.D, 16); }
  // frozen embeddings, DIN, D = 16, T <= 64: the att0 gradient's per-sample terms come out of this kernel's tail
  // (ChainX3Args::ab_*), launch_backward skips attn_bwd (GOCTR_CHAIN_ATTN_BWD=0: the separate kernel)
  m->attn_bwd_in_chain = o.train && c.kind == GOCTR_DIN && src.id_mode && !emb_plan_active(m) && c.D == 16 && c.T <= 64 &&
                         env_int("GOCTR_CHAIN_ATTN_BWD", 1) != 0;
  if (m->attn_bwd_in_chain) {
    a.ab_ids = src.ub_ids; a.ab_emb = src.emb; a.ab_V = src.V; a.ab_gate = fb.gate; a.ab_wgt = fb.wgt; a.ab_out = m->attp.p;
    a.ab_fac = gate_fac_mode(m, src, o, B) ? fb.fac : nullptr;
    a.ab_T = c.T; a.ab_Tp = m->Tp;
  }
  a.yhat = fb.yhat; a.lossrow = m->lossrow.p;
  a.xcd_affine = env_int("GOCTR_XCD_AFFINE", 1);
  // per-tile sums of dW2 and of the att0 terms instead of their operands -- where the wide weight-gradient launch follows (it
  // adds the tiles up; GOCTR_CHAIN_TILE_SUMS=0: the operands are stored and multiplied there, as until round 5)
  const bool tile_sums = o.train && dw_wide_path(m, B) && env_int("GOCTR_CHAIN_TILE_SUMS", 1) != 0;
  m->dw2_from_chain = tile_sums;
  m->att0_from_chain = tile_sums && m->attn_bwd_in_chain;
  a.tile_dw2 = m->dw2_from_chain ? m->tile_dw2.p : nullptr;
  a.tile_att0 = m->att0_from_chain ? m->tile_att0.p : nullptr;
  static DevBuf<unsigned long long> dbgbuf;
  const bool dbg = dbg_on("chain") && (hipStream_t)e.active == e.stream;   // (not from a serving slot)
  if (dbg && !dbgbuf.p && dbgbuf.alloc(4096)) return -1;
  a.dbg = dbg ? dbgbuf.p : nullptr;
  ProfScope ps(GOCTR_K_CHAIN);
  // (forward only: persistent workgroups walk the row tiles -- one workgroup per tile measured 34.8 against 33.4 us per 32 768 rows)
  const int ntiles = (int)cdiv(B, 32);
  const bool persist = !o.train && e.compute_units > 0;
  // forward only: four wavefronts per tile and two workgroups per CU (ctr_fwd4.h; GOCTR_FWD4=0: the 8-wavefront kernel)
  // (a launch of at most one tile per CU keeps the 8-wavefront kernel: 470 against 459 M rows/s at 256 tiles per launch; 512 tiles 551 -> 575 M,
  // 1024 tiles 608 -> 637 M -- profiles/r06_fwd4.txt)
  const bool fwd4 = !o.train && ntiles > e.compute_units && env_int("GOCTR_FWD4", 1) != 0;
  const dim3 grid((unsigned)(persist ? std::min(ntiles, (fwd4 ? 2 : 1) * e.compute_units) : ntiles));
  static const char* const kSym[3][2] = {{"ctr_chain_x3_kernel<2,false>", "ctr_chain_x3_kernel<2,true>"},
                                         {"ctr_chain_x3_kernel<9,false>", "ctr_chain_x3_kernel<9,true>"},
                                         {"ctr_chain_x3_kernel<15,false>", "ctr_chain_x3_kernel<15,true>"}};
  // (Round 4 also built a 16-row tile kernel -- two workgroups per CU -- which lost, 25.8 against 20.9 us at cfg3: a 16-row
  // tile's dependent pipeline is as long as a 32-row tile's.  The kernel left the tree in round 5; DESIGN_HISTORY.md and
  // profiles/r04_chain_x16_ab.txt keep the record, git keeps csrc/ctr_chain_x16.h.)
  if (ps.on) prof_note_kernel(GOCTR_K_CHAIN, fwd4 ? (m->x3_nch0 == 2 ? "ctr_fwd4_kernel<2,false>" : m->x3_nch0 == 9 ? "ctr_fwd4_kernel<9,false>" : "ctr_fwd4_kernel<15,true>")
                                                  : kSym[m->x3_nch0 == 2 ? 0 : m->x3_nch0 == 9 ? 1 : 2][o.train ? 0 : 1]);
  if (fwd4) launch_fwd4(m->x3_nch0, a, grid, e.active);
  else
  switch (m->x3_nch0) {
    case 2: launch_chain_x3_n<2>(a, grid, e.active, !o.train); break;
    case 9: launch_chain_x3_n<9>(a, grid, e.active, !o.train); break;
    default: launch_chain_x3_n<15>(a, grid, e.active, !o.train); break;
  }
  GOCTR_HIP(hipGetLastError());
  if (dbg) {
    unsigned long long h[CX_NSTAMP];
    if (dbgbuf.download(h, CX_NSTAMP)) return -1;
    if (fwd4) {
      fprintf(stderr, "fwd4 phases (s_memtime ticks): h0 split+barrier %lld | F0 %lld | epi0+F1 %lld | xchg barrier %lld | epi1+z2 %lld | total %lld\n",
              (long long)(h[1] - h[0]), (long long)(h[2] - h[1]), (long long)(h[3] - h[2]), (long long)(h[4] - h[3]), (long long)(h[5] - h[4]),
              (long long)(h[5] - h[0]));
      fprintf(stderr, "fwd4 workgroup 0: %lld shader cycles in %.2f us (100 MHz clock) = %.2f GHz\n", (long long)(h[5] - h[6]),
              (double)(h[8] - h[7]) * 0.01, (double)(h[5] - h[6]) / ((double)(h[8] - h[7]) * 10.0));
      // every workgroup's lifetime (100 MHz clock) and where it ran
      std::vector<unsigned long long> all(4096);
      if (dbgbuf.download(all.data(), 4096)) return -1;
      std::vector<double> life; std::vector<unsigned long long> where;
      for (unsigned b = 0; b < grid.x && b < 960; ++b) {
        life.push_back((double)(all[128 + 4 * b + 1] - all[128 + 4 * b]) * 0.01);
        where.push_back((all[128 + 4 * b + 3] & 0xf) << 16 | (all[128 + 4 * b + 2] & 0xff00));      // XCC | SE, SH, CU of HW_ID
      }
      std::sort(life.begin(), life.end()); std::sort(where.begin(), where.end());
      const size_t cus = (size_t)(std::unique(where.begin(), where.end()) - where.begin());
      if (!life.empty())
        fprintf(stderr, "fwd4 workgroups: %zu on %zu CUs, lifetimes min %.2f / median %.2f / max %.2f us\n", life.size(), cus, life.front(),
                life[life.size() / 2], life.back());
    } else if (!o.train) {
      fprintf(stderr, "chain_x3 forward-only phases (s_memtime ticks): h0 split+barrier %lld | F0 %lld | epi0 %lld | F1 %lld | xchg barrier %lld | epi1+z2 %lld | total %lld\n",
              (long long)(h[1] - h[0]), (long long)(h[2] - h[1]), (long long)(h[3] - h[2]), (long long)(h[4] - h[3]), (long long)(h[5] - h[4]),
              (long long)(h[6] - h[5]), (long long)(h[6] - h[0]));
      fprintf(stderr, "chain_x3 forward-only workgroup 0: %lld shader cycles in %.2f us (100 MHz clock) = %.2f GHz\n", (long long)(h[6] - h[14]),
              (double)(h[13] - h[15]) * 0.01, (double)(h[6] - h[14]) / ((double)(h[13] - h[15]) * 10.0));
    } else {
      fprintf(stderr, "chain_x3 phases (s_memtime ticks): h0 split+barrier %lld | F0(+epi tile0) %lld | F1(+epi tile1) %lld | xchg barrier %lld | epi1+z2+dz1 %lld | B0(+epi) %lld | dp %lld + xchg %lld | total %lld\n",
              (long long)(h[1] - h[0]), (long long)(h[2] - h[1]), (long long)(h[4] - h[2]), (long long)(h[5] - h[4]), (long long)(h[6] - h[5]),
              (long long)(h[7] - h[6]), (long long)(h[8] - h[7]), (long long)(h[9] > h[8] ? h[9] - h[8] : 0),
              (long long)((h[9] > h[8] ? h[9] : h[8]) - h[0]));
    }
  }
  return 0;
}

ChainArgs make_chain_args(goctr_model* m, const RowSource& src, int B, const StepOpts& o, const StepState* st, const FwdBufs& fb) {
  const goctr_ctr_cfg& c = m->cfg;
  Engine& e = engine();
  const uint32_t row_off = (uint32_t)(e.eff_rank() * B);
  const bool drop = o.train && o.drop_mode != 0;
  ChainArgs a{};
  a.h0 = fb.h0; a.Ip = m->Ip;
  a.W0i = m->img(0); a.W1i = m->img(1); a.W1Ti = m->img(2); a.W0sTi = m->img(3); a.w2 = m->W2T.p;
  a.H1 = c.H1; a.H2 = c.H2; a.H1p = m->H1p; a.H2p = m->H2p; a.Dp = m->Dp; a.B = B;
  a.train = o.train ? 1 : 0; a.kind = c.kind;
  a.d0 = DropCfg{drop && o.p0 > 0 ? o.drop_mode : 0, o.p0, m->mask0.p, c.H1, o.seed, 0u, row_off};
  a.d1 = DropCfg{drop && o.p1 > 0 ? o.drop_mode : 0, o.p1, m->mask1.p, c.H2, o.seed, 1u, row_off};
  a.st = st; a.Y = src.Y; a.rows = src.rows; a.inv_bglobal = 1.0f / (float)(B * e.eff_world());
  a.buf_floats = chain_buf_floats(m->Ip, m->H1p, m->H2p);
  a.A0 = m->A0.p; a.A1 = m->A1.p; a.dz0 = m->dz0.p; a.dz1 = m->dz1.p; a.dz2 = m->dz2.p; a.dp = m->dp.p;   // (forward only: none of these is touched)
  a.yhat = fb.yhat; a.lossrow = m->lossrow.p;
  return a;
}

AttnArgs make_attn_args(goctr_model* m, const RowSource& src, int B, const StepState* st, const FwdBufs& fb, bool fac);
int attn_fast_mode(const goctr_model* m, const RowSource& src, int* groups);
// A small serving pass in key mode as ONE launch (ctr_serve.h): the shapes with a compile-time attention variant at 8, 16
// or 64 embedding columns, launches the 16-row forward kernel would take (too few rows for a 32-row tile per CU)
bool serve16_ok(const goctr_model* m, const RowSource& src, int B) {
  int groups = 0;
  const int fast = attn_fast_mode(m, src, &groups);
  StepOpts o; o.train = false;
  return src.k_users && fast != 0 && (groups == 2 || groups == 4 || groups == 16) && chain_ok(m) && !chain_x3_ok(m, o, B) &&
         cdiv(B, 32) < engine().compute_units && env_int("GOCTR_SERVE_ONE_LAUNCH", 1) != 0;
}
int launch_serve16(goctr_model* m, const RowSource& src, int B, const StepState* st, const FwdBufs& fb, unsigned* done, unsigned epoch) {
  int groups = 0;
  const int fast = attn_fast_mode(m, src, &groups);
  StepOpts o; o.train = false;
  ChainArgs a = make_chain_args(m, src, B, o, st, fb);
  a.done = done; a.epoch = epoch;
  AttnArgs aa = make_attn_args(m, src, B, st, fb, false);
  aa.gate = nullptr; aa.wgt = nullptr;                  // (only the training step's backward reads gates and weights)
  const size_t lds = chain_lds_bytes<5>(m->Ip, m->H1p, m->H2p);
  const dim3 grid((unsigned)cdiv(B, 16)), blk(1024);
  hipStream_t s = engine().active;
#define GOCTR_SERVE16_H(L, H)                                                                          \
  do {                                                                                                 \
    if (fast == 1) hipLaunchKernelGGL((ctr_serve16_kernel<L, 1, H>), grid, blk, lds, s, aa, a);        \
    else if (fast == 2) hipLaunchKernelGGL((ctr_serve16_kernel<L, 2, H>), grid, blk, lds, s, aa, a);   \
    else hipLaunchKernelGGL((ctr_serve16_kernel<L, 3, H>), grid, blk, lds, s, aa, a);                  \
  } while (0)
#define GOCTR_SERVE16(L) do { if (m->Ip <= 160) GOCTR_SERVE16_H(L, 10); else GOCTR_SERVE16_H(L, 15); } while (0)
  if (groups == 2) GOCTR_SERVE16(2);
  else if (groups == 4) GOCTR_SERVE16(4);
  else GOCTR_SERVE16(16);
#undef GOCTR_SERVE16_H
#undef GOCTR_SERVE16
  GOCTR_HIP(hipGetLastError());
  return 0;
}

int launch_chain(goctr_model* m, const RowSource& src, int B, const StepOpts& o, const StepState* st, const FwdBufs& fb) {
  if (chain_x3_ok(m, o, B)) return launch_chain_x3(m, src, B, o, st, fb);
  Engine& e = engine();
  ChainArgs a = make_chain_args(m, src, B, o, st, fb);
  static DevBuf<unsigned long long> dbgbuf;
  const bool dbg = o.train && dbg_on("chain");
  if (dbg && !dbgbuf.p && dbgbuf.alloc(CHAIN_NSTAMP)) return -1;
  a.dbg = dbg ? dbgbuf.p : nullptr;
  ProfScope ps(GOCTR_K_CHAIN);
  const dim3 grid((unsigned)cdiv(B, 32));
  const size_t lds = chain_lds_bytes<5>(m->Ip, m->H1p, m->H2p);
  const int dmode = (a.d0.mode || a.d1.mode) ? o.drop_mode : 0;
  // forward only and too few rows to give every CU a 32-row workgroup: 16-row workgroups, H1 split over 4 wavefronts
  if (!o.train && cdiv(B, 32) < e.compute_units) {
    if (ps.on) prof_note_kernel(GOCTR_K_CHAIN, "ctr_fwd16_kernel<4,5>");
    hipLaunchKernelGGL((ctr_fwd16_kernel<4, 5>), dim3((unsigned)cdiv(B, 16)), dim3(512), lds, e.active, a);
  } else
  if (dmode == 0) { if (ps.on) prof_note_kernel(GOCTR_K_CHAIN, "ctr_chain_kernel<7,5,0>"); hipLaunchKernelGGL((ctr_chain_kernel<7, 5, 0>), grid, dim3(512), lds, e.active, a); }
  else if (dmode == 1) { if (ps.on) prof_note_kernel(GOCTR_K_CHAIN, "ctr_chain_kernel<7,5,1>"); hipLaunchKernelGGL((ctr_chain_kernel<7, 5, 1>), grid, dim3(512), lds, e.active, a); }
  else { if (ps.on) prof_note_kernel(GOCTR_K_CHAIN, "ctr_chain_kernel<7,5,2>"); hipLaunchKernelGGL((ctr_chain_kernel<7, 5, 2>), grid, dim3(512), lds, e.active, a); }
  GOCTR_HIP(hipGetLastError());
  if (dbg) {
    unsigned long long h[CHAIN_NSTAMP];
    if (dbgbuf.download(h, CHAIN_NSTAMP)) return -1;
    fprintf(stderr, "chain phases (s_memtime ticks):");
    for (int k = 1; k < 10; ++k) fprintf(stderr, " %d:%lld", k, (long long)(h[k] - h[k - 1]));
    fprintf(stderr, " | ph0 mma %lld bar %lld | ph1 mma %lld bar %lld | F1 mma+xw %lld bar %lld", (long long)(h[10] - h[1]), (long long)(h[2] - h[10]),
            (long long)(h[11] - h[2]), (long long)(h[3] - h[11]), (long long)(h[12] - h[4]), (long long)(h[13] - h[12]));
    fprintf(stderr, "  total %lld\n", (long long)(h[9] - h[0]));
  }
  return 0;
}

// forward part: kernels 1-4
// `par`: which copy of gate / wgt the launch writes (the parity of the step the gather belongs to)
FwdBufs train_bufs(goctr_model* m, int par) { return FwdBufs{m->h0.p, m->gate_p(par), m->wgt_p(par), m->yhat.p, m->P0.p, m->P1.p, m->gfac_p(par)}; }
// Round 6: where the ONLY reader of a training step's gates and similarity weights is the attention backward at the chain launch's
// tail (DIN, frozen embeddings, id mode, the bf16-split chain: launch_chain_x3's attn_bwd_in_chain, which this predicate implies), the
// attention forward leaves the one factor (g (1 - g)) w that backward multiplies with (AttnArgs::fac) instead of the two arrays.  The
// producer -- the step's own attention launch, or the previous step's last launch -- and the consumer evaluate this with the same
// (model, rows, options, batch); a start carried over from another call compares H0Carry::fac.  GOCTR_GATE_FAC=0: both arrays.
bool gate_fac_mode(const goctr_model* m, const RowSource& src, const StepOpts& o, int B) {
  const goctr_ctr_cfg& c = m->cfg;
  return o.train && c.kind == GOCTR_DIN && src.id_mode && m->emb_lr <= 0.f && c.D == 16 && c.T <= 64 && chain_ok(m) &&
         chain_x3_ok(m, o, B) && env_int("GOCTR_CHAIN_ATTN_BWD", 1) != 0 && env_int("GOCTR_GATE_FAC", 1) != 0;
}
// fac: gate_fac_mode() of the step that will consume the launch's rows
AttnArgs make_attn_args(goctr_model* m, const RowSource& src, int B, const StepState* st, const FwdBufs& fb, bool fac) {
  const goctr_ctr_cfg& c = m->cfg;
  AttnArgs aa{};
  aa.src = src; aa.st = st; aa.B = B; aa.U = c.U; aa.T = c.T; aa.D = c.D; aa.C = c.C; aa.Ip = m->Ip;
  aa.kind = c.kind; aa.att = c.att; aa.att0 = m->W.p + m->offa; aa.h0 = fb.h0; aa.gate = fb.gate; aa.wgt = fb.wgt;
  aa.Tp_att = m->Tp;
  aa.xcd_affine = env_int("GOCTR_XCD_AFFINE", 1);      // (ctr_kernels.h xcd_unit_of_block; a permutation of the workgroups' samples)
  aa.inv_T = 1.0f / (float)c.T;
  if (fac && fb.fac) { aa.fac = fb.fac; aa.gate = nullptr; aa.wgt = nullptr; }
  return aa;
}
AttnArgs make_attn_args(goctr_model* m, const RowSource& src, int B, const StepState* st, int par, bool fac) {
  return make_attn_args(m, src, B, st, train_bufs(m, par), fac);
}

// the compile-time mode launch_attn_fwd picks for this model's rows, or 0; `groups` = lanes per embedding row
int attn_fast_mode(const goctr_model* m, const RowSource& src, int* groups) {
  const goctr_ctr_cfg& c = m->cfg;
  const bool vec4 = src.id_mode && c.D % 4 == 0;
  const int g = vec4 ? c.D / 4 : c.D;
  if (groups) *groups = g;
  const bool small_table = (unsigned long long)(src.V + 1) * (unsigned long long)c.D * 4ull < (1ull << 32);   // (32-bit row offsets in those kernels)
  return !(vec4 && small_table && g * 4 == c.D && (g & (g - 1)) == 0) ? 0 : c.kind != GOCTR_DIN ? 1 : (c.att == GOCTR_ATT_COSINE ? 2 : 3);
}
// (Round 5 tried the next batch's attention on a second stream BESIDE the weight-gradient and reduce launches instead of inside
// the step's last launch: it lost, 58.5 against 46.2 us per cfg3 step -- the two branches slow each other down by what they were
// to hide, and a cross-stream edge in a captured graph costs ~6 us here; profiles/r05_fork_ab.txt, commits 0af81b4 .. ae775f3.)
// can the steps of a graph be pipelined (reduce_attn_kernel)?  Single GPU, fused update, the fused chain, D = 16 or 64 rows
bool pipeline_ok(const goctr_model* m, const RowSource& src) {
  int groups = 0;
  const int fast = attn_fast_mode(m, src, &groups);
  // (DIN: one reduce block must own the whole att0 segment -- it publishes the flag the attention workgroups wait for)
  const bool one_block = m->cfg.kind != GOCTR_DIN || (m->offa * 2) / 256 == ((m->offa + m->Tp) * 2 - 1) / 256;
  if (engine().comm_active()) {
    // data parallel (dense all-reduce only): the part behind the collective -- Adam -- shares its launch with the next step's
    // attention (adam_attn_kernel); one 256-parameter Adam block must own the att0 segment
    const bool one_adam_block = m->cfg.kind != GOCTR_DIN || m->offa / 256 == (m->offa + m->Tp - 1) / 256;
    return fast != 0 && (groups == 4 || groups == 16) && one_adam_block && chain_ok(m) && m->emb_lr <= 0.f &&
           env_int("GOCTR_PIPELINE", 1) != 0;
  }
  return fast != 0 && (groups == 4 || groups == 16) && one_block && chain_ok(m) &&
         env_int("GOCTR_PIPELINE", 1) != 0;
}

int launch_reduce_attn(goctr_model* m, const RowSource& src, int B, const StepOpts& o, const ReduceAdamArgs& p) {
  int groups = 0;
  const int fast = attn_fast_mode(m, src, &groups);
  const AttnArgs aa = make_attn_args(m, src, B, p.r.st, m->stp ^ 1, gate_fac_mode(m, src, o, B));      // the NEXT step's gates
  const int nred = (int)cdiv((int64_t)m->nflat * 2, 256) + 1;
  const dim3 grid((unsigned)(nred + cdiv(B, 4))), blk(256);
  hipStream_t st = engine().active;
#define GOCTR_RA(L)                                                                                        \
  do {                                                                                                     \
    if (fast == 1) hipLaunchKernelGGL((reduce_attn_kernel<4, L, 1>), grid, blk, 0, st, p, aa, nred);       \
    else if (fast == 2) hipLaunchKernelGGL((reduce_attn_kernel<4, L, 2>), grid, blk, 0, st, p, aa, nred);  \
    else hipLaunchKernelGGL((reduce_attn_kernel<4, L, 3>), grid, blk, 0, st, p, aa, nred);                 \
  } while (0)
  if (groups == 4) GOCTR_RA(4);
  else GOCTR_RA(16);
#undef GOCTR_RA
  GOCTR_HIP(hipGetLastError());
  return 0;
}

// fbp: where a forward-only pass keeps its rows (null: the training workspace)
int launch_forward(goctr_model* m, const RowSource& src, int B, const StepOpts& o, const StepState* st_override = nullptr,
                   const FwdBufs* fbp = nullptr) {
  const goctr_ctr_cfg& c = m->cfg;
  Engine& e = engine();
  const StepState* st = st_override ? st_override : m->st_cur();
  const FwdBufs fb = fbp ? *fbp : train_bufs(m, m->stp);
  if (!o.pipelined) {
    AttnArgs aa = make_attn_args(m, src, B, st, fb, gate_fac_mode(m, src, o, B));
    if (!o.train) { aa.gate = nullptr; aa.wgt = nullptr; }     // (only the backward reads them: 13 MB less per 32 768-row launch)
    if (launch_attn_fwd(aa)) return -1;
  }
  if (o.train) { m->dpv_from_chain = false; m->attn_bwd_in_chain = false; m->dw2_from_chain = false; m->att0_from_chain = false; m->att0_early = false; }   // (launch_chain_x3 / launch_backward set them when they do the work themselves)
  if (chain_ok(m)) return launch_chain(m, src, B, o, st, fb);  // layers + (when training) backward-data, fused

  const int bglobal = B * e.eff_world();
  const uint32_t row_off = (uint32_t)(e.eff_rank() * B);
  const bool drop = o.train && o.drop_mode != 0;
  DropCfg d0{drop && o.p0 > 0 ? o.drop_mode : 0, o.p0, m->mask0.p, c.H1, o.seed, 0u, row_off};
  DropCfg d1{drop && o.p1 > 0 ? o.drop_mode : 0, o.p1, m->mask1.p, c.H2, o.seed, 1u, row_off};
  // (forward only: no dropout, so the post-dropout copies A0 / A1 are not written)
  EpiSigDrop e0{fb.P0, d0.mode ? m->A0.p : fb.P0, m->H1p, c.H1, d0, st};
  if (launch_nn(GOCTR_K_GEMM_FWD0, fb.h0, m->Ip, m->W.p, m->H1p, B, m->Ip, m->H1p, e0)) return -1;
  const float* A0 = d0.mode ? m->A0.p : fb.P0;
  EpiSigDrop e1{fb.P1, d1.mode ? m->A1.p : fb.P1, m->H2p, c.H2, d1, st};
  if (launch_nn(GOCTR_K_GEMM_FWD1, A0, m->H1p, m->W.p + m->off1, m->H2p, B, m->H1p, m->H2p, e1)) return -1;
  const float* A1 = d1.mode ? m->A1.p : fb.P1;
  EpiOut eo{fb.yhat, o.train ? m->lossrow.p : nullptr, o.train ? m->dz2.p : nullptr, src.Y, src.rows, st, B,
            1.0f / (float)bglobal};
  if (launch_nn(GOCTR_K_GEMM_OUT, A1, m->H2p, m->W.p + m->off2, 16, B, m->H2p, 16, eo)) return -1;
  return 0;
}

// backward part up to and including the slab reduce: kernels 5-12
AdamArgs make_adam_args(goctr_model* m, int B, const goctr_train_cfg& tc);

// Buffers of the sparse embedding update.  Allocated (and zeroed) BEFORE a step is captured into a hipGraph: a
// hipMemsetAsync issued during capture becomes a graph node and would re-zero hundreds of MB on every replay.
int ensure_emb_workspace(goctr_model* m, long long V, int B) {
  // (the accumulators are sized for this rank's own ids; a communicator created after the first step changes the index space)
  if (m->emb_lr <= 0.f || (m->emb_V == V && m->emb_B == B && m->emb_world == engine().eff_world() &&
                           m->emb_comm == engine().comm_active())) return 0;
  const goctr_ctr_cfg& c = m->cfg;
  const int Np = round_up(2 * c.D, 16);
  const int W = engine().comm_active() ? engine().eff_world() : 1;
  const long long Vw = round_up((int)cdiv(V, W), 4);
  const long long Vp = Vw * W;                                        // owner-major index space (emb_train.h: emb_pidx)
  const long long cap = std::min<long long>(V, (long long)B * (c.T + 1));
  if (m->dpv.alloc((size_t)B * Np) || m->W0pvT.alloc((size_t)m->H1p * Np) || m->emb_mark.alloc((size_t)Vp) ||
      m->emb_rank.alloc((size_t)Vp, false) || m->emb_total.alloc(1) || m->emb_accum.alloc((size_t)cap * c.D) ||
      m->emb_slot_id.alloc((size_t)cap, false) || m->emb_tiles.alloc((size_t)cdiv(Vp, SCAN_TILE), false))
    return -1;
  if (engine().comm_active()) {
    if (m->ex_off.alloc(W + 1) || m->ex_cnt.alloc(W) || m->ex_allcnt.alloc((size_t)W * W) || m->ex_nred.alloc(1) ||
        m->ex_allnred.alloc(W) || m->ex_red_total.alloc(1)) return -1;
    // (the data buffers grow on demand: their sizes follow the ids the batches actually touch)
  }
  GOCTR_HIP(hipStreamSynchronize(engine().stream));
  m->emb_V = V; m->emb_B = B; m->emb_world = engine().eff_world(); m->emb_comm = engine().comm_active(); m->emb_Vw = Vw;
  m->w0pv_live = false; m->plan.valid = false;      // (W0pvT was reallocated; the plan's index space may have changed)
  m->graph.destroy();
  return 0;
}

template <int GS, bool CACHE>
void launch_emb_grad2(int mode, dim3 gb, size_t lds, hipStream_t s, const EmbTrainArgs& a, int nslot) {
  if (mode == 0) hipLaunchKernelGGL((emb_grad_kernel<GS, 0, CACHE>), gb, dim3(EMB_GRAD_THREADS), lds, s, a, nslot);
  else if (mode == 1) hipLaunchKernelGGL((emb_grad_kernel<GS, 1, CACHE>), gb, dim3(EMB_GRAD_THREADS), lds, s, a, nslot);
  else hipLaunchKernelGGL((emb_grad_kernel<GS, 2, CACHE>), gb, dim3(EMB_GRAD_THREADS), lds, s, a, nslot);
}
template <int GS>
void launch_emb_grad(int mode, bool cache, dim3 gb, size_t lds, hipStream_t s, const EmbTrainArgs& a, int nslot) {
  if (cache) launch_emb_grad2<GS, true>(mode, gb, lds, s, a, nslot);
  else launch_emb_grad2<GS, false>(mode, gb, 0, s, a, nslot);
}

// Sparse embedding update of one step (emb_train.h).  Runs after every reader of the table in this step (attn_fwd,
// attn_bwd's re-gather) and before the step state advances.
// Bucketed exchange of one step's sparse row gradients (emb_train.h): all-to-all of the (id,
// fixed-point row) pairs to their owners (id % world), exact owner-side sums, all-gather of (id, delta), every replica
// applies every delta.  Two small host read-backs size the transfers (the counts are data dependent), so these steps run
// eagerly; the traffic is proportional to the ids the batches touch, not to the vocabulary.
int launch_emb_exchange(goctr_model* m, const EmbTrainArgs& a) {
  Engine& e = engine();
  const int W = e.eff_world(), r = e.eff_rank(), D = a.D;
  hipStream_t s = e.stream;
  hipLaunchKernelGGL(emb_bucket_bounds_kernel, dim3(1), dim3(64), 0, s, m->emb_slot_id.p, m->emb_total.p, W, m->ex_off.p, m->ex_cnt.p);
  GOCTR_HIP(hipGetLastError());
  if (comm_allgather_i32(m->ex_cnt.p, m->ex_allcnt.p, (size_t)W)) return -1;
  std::vector<int> off(W + 1), allcnt((size_t)W * W);
  if (m->ex_off.download(off.data(), W + 1) || m->ex_allcnt.download(allcnt.data(), (size_t)W * W)) return -1;   // (host sync 1)
  std::vector<size_t> so(W), sc(W), ro(W), rc(W);
  size_t nrecv = 0;
  for (int p = 0; p < W; ++p) {
    so[p] = (size_t)off[p]; sc[p] = (size_t)(off[p + 1] - off[p]);
    ro[p] = nrecv; rc[p] = (size_t)allcnt[(size_t)p * W + r]; nrecv += rc[p];
  }
  if (m->ex_rids.ensure(nrecv, false) || m->ex_rrows.ensure(nrecv * D, false)) return -1;
  if (comm_alltoallv(m->emb_slot_id.p, so.data(), sc.data(), m->ex_rids.p, ro.data(), rc.data(), 4)) return -1;
  std::vector<size_t> soD(W), scD(W), roD(W), rcD(W);
  for (int p = 0; p < W; ++p) { soD[p] = so[p] * D; scD[p] = sc[p] * D; roD[p] = ro[p] * D; rcD[p] = rc[p] * D; }
  if (comm_alltoallv(m->emb_accum.p, soD.data(), scD.data(), m->ex_rrows.p, roD.data(), rcD.data(), 8)) return -1;
  double sent = 0;
  for (int p = 0; p < W; ++p) sent += (double)sc[p] * (4 + 8.0 * D);
  // the local accumulators are done with (sent): clear them for the next step
  GOCTR_HIP(hipMemsetAsync(m->emb_accum.p, 0, sizeof(long long) * (size_t)off[W] * D, s));
  // owner side: unique ids of my bucket -> dense slots (ascending id), exact sums
  const size_t cap_red = std::min<size_t>((size_t)m->emb_Vw, nrecv);
  if (m->ex_red.n < cap_red * D || !m->ex_red.p) { if (m->ex_red.alloc(std::max<size_t>(cap_red * D, 1))) return -1; }   // (zeroed; kept zero by emb_delta)
  if (m->ex_red_ids.ensure(std::max<size_t>(cap_red, 1), false) || m->ex_delta.ensure(std::max<size_t>(cap_red * D, 1), false)) return -1;
  if (nrecv) {
    hipLaunchKernelGGL(emb_recv_mark_kernel, dim3((unsigned)cdiv((long long)nrecv, 256)), dim3(256), 0, s, m->ex_rids.p, (long long)nrecv, W,
                       m->emb_Vw, m->emb_mark.p);
    GOCTR_HIP(hipGetLastError());
  }
  if (exclusive_scan_sink(m->emb_mark.p + (size_t)r * m->emb_Vw, m->emb_Vw, m->emb_tiles, m->ex_red_total.p, EmbMultiMap{},
                          EmbRankSink{m->emb_mark.p, m->emb_rank.p, m->ex_red_ids.p, W, m->emb_Vw, (long long)r * m->emb_Vw})) return -1;
  if (nrecv) {
    hipLaunchKernelGGL(emb_recv_accumulate_kernel, dim3((unsigned)cdiv((long long)nrecv * D, 256)), dim3(256), 0, s, m->ex_rids.p,
                       m->ex_rrows.p, (long long)nrecv, D, W, m->emb_Vw, m->emb_rank.p, m->ex_red.p);
    GOCTR_HIP(hipGetLastError());
  }
  const int cus = e.compute_units > 0 ? e.compute_units : 256;
  hipLaunchKernelGGL(emb_delta_kernel, dim3((unsigned)std::min<long long>(std::max<long long>(cdiv((long long)cap_red * D, 256), 1), 16 * cus)),
                     dim3(256), 0, s, m->ex_red.p, m->ex_red_total.p, D, a.lr, m->ex_delta.p);
  GOCTR_HIP(hipGetLastError());
  // all-gather of (ids, deltas): counts first
  hipLaunchKernelGGL(emb_count_to_i32_kernel, dim3(1), dim3(1), 0, s, m->ex_red_total.p, m->ex_nred.p);
  GOCTR_HIP(hipGetLastError());
  if (comm_allgather_i32(m->ex_nred.p, m->ex_allnred.p, 1)) return -1;
  std::vector<int> nred(W);
  if (m->ex_allnred.download(nred.data(), W)) return -1;                                                           // (host sync 2)
  std::vector<size_t> go(W), gc(W), zo(W, 0), mine(W);
  size_t ng = 0;
  for (int p = 0; p < W; ++p) { go[p] = ng; gc[p] = (size_t)nred[p]; ng += gc[p]; mine[p] = (size_t)nred[r]; }
  if (m->ex_gids.ensure(std::max<size_t>(ng, 1), false) || m->ex_gdelta.ensure(std::max<size_t>(ng * D, 1), false)) return -1;
  if (comm_alltoallv(m->ex_red_ids.p, zo.data(), mine.data(), m->ex_gids.p, go.data(), gc.data(), 4)) return -1;
  std::vector<size_t> goD(W), gcD(W), mineD(W);
  for (int p = 0; p < W; ++p) { goD[p] = go[p] * D; gcD[p] = gc[p] * D; mineD[p] = mine[p] * D; }
  if (comm_alltoallv(m->ex_delta.p, zo.data(), mineD.data(), m->ex_gdelta.p, goD.data(), gcD.data(), 4)) return -1;
  sent += (double)W * nred[r] * (4 + 4.0 * D);
  m->ex_bytes_last = sent;
  if (ng) {
    hipLaunchKernelGGL(emb_apply_gathered_kernel, dim3((unsigned)std::min<long long>(cdiv((long long)ng * D, 256), 16 * cus)), dim3(256), 0, s,
                       a.emb, m->ex_gids.p, m->ex_gdelta.p, (long long)ng, D);
    GOCTR_HIP(hipGetLastError());
  }
  return 0;
}

// Shapes the id-major plan path covers (emb_train.h "Round 3"); everything else keeps emb_grad_kernel's atomics.
bool emb_plan_ok(const goctr_model* m, int B) {
  const goctr_ctr_cfg& c = m->cfg;
  const bool lay = c.kind != GOCTR_DIN || c.D == 4 || c.D == 8 || c.D == 16 || c.D == 32 || c.D == 64;
  return lay && c.D <= 64 && c.T < (1 << EMB_PAIR_TBITS) && B < (1 << (31 - EMB_PAIR_TBITS)) && env_int("GOCTR_EMB_PLAN", 1) != 0;
}

// The plan is resident for the whole dataset (12 B per pair + 8 B per slot): bounded by GOCTR_EMB_PLAN_MAX_MB (default 32 768,
// of 288 GB); a dataset beyond it keeps the atomics path (emb_grad_kernel), with a note on stderr, instead of failing an
// allocation deep inside the first step.
bool emb_plan_fits(const goctr_model* m, const goctr_dataset* d, long long V, int B) {
  const long long per = m->cfg.T + 1, nb = cdiv(d->rows, B);
  const double bytes = 12.0 * (double)(nb * B * per) + 8.0 * (double)(nb * std::min<long long>((long long)B * per, V));
  const double budget = (double)env_int("GOCTR_EMB_PLAN_MAX_MB", 32768) * 1048576.0;
  if (bytes <= budget) return true;
  static std::atomic<bool> said{false};
  if (!said.exchange(true))
    fprintf(stderr, "goctr: the sparse plan of this dataset would take %.1f GB (> GOCTR_EMB_PLAN_MAX_MB = %d): embedding training "
            "uses the atomics path\n", bytes / 1073741824.0, env_int("GOCTR_EMB_PLAN_MAX_MB", 32768));
  return false;
}

// Build (or reuse) the sparse plan of dataset d at batch size B: per batch the distinct ids in ascending owner-major order
// and the (sample, slot) pairs sorted by id.  One-time work per dataset, outside every capture: a count per id, two prefix
// sums over the vocabulary and a fill per batch, with one small read-back per batch to advance the bases.
int ensure_emb_plan(goctr_model* m, const goctr_dataset* d, const RowSource& src, int B) {
  Engine& e = engine();
  const goctr_ctr_cfg& c = m->cfg;
  const int W = e.comm_active() ? e.eff_world() : 1;
  auto& P = m->plan;
  if (P.valid && P.ds == d->uid && P.V == src.V && P.B == B && P.W == W && P.T == c.T) return 0;
  P.valid = false;
  hipStream_t s = e.stream;
  GOCTR_HIP(hipStreamSynchronize(s));
  m->graph.destroy();                                  // captured launches bake the plan's pointers in
  const long long Vw = m->emb_Vw;
  const long long nb = cdiv(d->rows, B), per = c.T + 1;
  // (emb_plan.hip: a stable sort of each batch's keys by owner-major row + one flag / scan / fill pass; no atomics, no
  // per-batch read-back, temporaries sized for one batch)
  const size_t np_cap = (size_t)(nb * B * per), ns_cap = (size_t)(nb * std::min<long long>((long long)B * per, src.V));
  if (P.pair.alloc(np_cap, false) || P.pslot.alloc(np_cap, false) || P.pid.alloc(np_cap, false) || P.slot_id.alloc(ns_cap, false) ||
      P.slot_off.alloc(ns_cap + (size_t)nb, false) || P.pair_off.alloc((size_t)nb + 1, false) || P.slot_base.alloc((size_t)nb + 1, false)) return -1;
  long long tot[4] = {0, 0, 0, 0};
  const auto t_build = std::chrono::steady_clock::now();
  {
    ProfScope ps(GOCTR_K_EMB_PLAN);
    if (emb_plan_build(EmbPlanSource{src.ub_ids, src.item_ids, src.rows, src.V}, B, c.T, W, Vw, nb,
                       EmbPlanArrays{P.pair.p, P.pslot.p, P.pid.p, P.slot_id.p, P.slot_off.p, P.pair_off.p, P.slot_base.p}, tot)) return -1;
  }
  const long long max_pairs = tot[2], max_slots = tot[3];
  P.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();
  if (c.kind == GOCTR_DIN && (m->emb_dx.ensure((size_t)B * c.T * c.D, false) || m->emb_gsum.ensure((size_t)B * c.D, false))) return -1;
  m->ex_fixed = false;
  if (e.comm_active() && env_int("GOCTR_EMB_FIXED_EXCHANGE", 1) != 0) {
    // bucket bounds of every batch, the largest bucket over batches, owners AND ranks (one small all-gather, here, once)
    GOCTR_CHECK(W <= 1023, "world %d too large for the bucket kernel", W);
    if (m->ex_bucket_off.alloc((size_t)nb * (W + 1), false)) return -1;
    hipLaunchKernelGGL(emb_plan_buckets_kernel, dim3((unsigned)nb), dim3((unsigned)round_up(W + 1, 64)), 0, s, P.view(), nb, W, m->ex_bucket_off.p);
    GOCTR_HIP(hipGetLastError());
    std::vector<int> boff((size_t)nb * (W + 1));
    if (m->ex_bucket_off.download(boff.data(), boff.size())) return -1;
    int smax = 1;
    for (long long k = 0; k < nb; ++k)
      for (int o = 0; o < W; ++o) smax = std::max(smax, boff[(size_t)k * (W + 1) + o + 1] - boff[(size_t)k * (W + 1) + o]);
    DevBuf<int> one, all;
    if (one.alloc(1, false) || all.alloc((size_t)W, false) || one.upload(&smax, 1)) return -1;
    if (comm_allgather_i32(one.p, all.p, 1)) return -1;
    std::vector<int> hs((size_t)W);
    if (all.download(hs.data(), (size_t)W)) return -1;
    for (int v : hs) smax = std::max(smax, v);
    const int S = round_up(smax, 4);
    const long long R = std::min<long long>(Vw, (long long)W * S);
    m->ex_S = S; m->ex_R = (int)R;
    const size_t ws = (size_t)W * S, wr = (size_t)W * (size_t)R;
    if (m->ex_send_ids.alloc(ws, false) || m->ex_recv_ids.alloc(ws, false) || m->ex_send_rows.alloc(ws * c.D, false) ||
        m->ex_recv_rows.alloc(ws * c.D, false) || m->ex_red.alloc(std::max<size_t>((size_t)R * c.D, 1)) ||     // (zeroed; kept zero by emb_delta)
        m->ex_red_ids.alloc(std::max<size_t>((size_t)R, 1), false) || m->ex_delta.alloc(std::max<size_t>((size_t)R * c.D, 1), false) ||
        m->ex_gids.alloc(std::max<size_t>(wr, 1), false) || m->ex_gdelta.alloc(std::max<size_t>(wr * c.D, 1), false)) return -1;
    GOCTR_HIP(hipStreamSynchronize(s));
    m->ex_fixed = true;
    // bytes this rank sends per step: W padded buckets of (id, fixed-point row) + its padded (id, delta) list to every rank
    m->ex_bytes_last = (double)W * S * (4 + 8.0 * c.D) + (double)W * (double)R * (4 + 4.0 * c.D);
  }
  P.ds = d->uid; P.V = src.V; P.B = B; P.W = W; P.T = c.T; P.nb = nb; P.max_pairs = max_pairs; P.max_slots = max_slots;
  P.total_pairs = tot[0]; P.total_slots = tot[1];
  P.valid = true;
  return 0;
}

template <int GS, int VEC>
int launch_emb_slot_gv(int mode, bool direct, long long max_pairs, hipStream_t s, const EmbSlotArgs& a) {
  const long long wgp = EmbSlotGeo<GS, VEC>::WGP;
  const dim3 grid((unsigned)std::max<long long>(cdiv(max_pairs, wgp), 1));
#define GOCTR_SLOT(M) do { if (direct) hipLaunchKernelGGL((emb_slot_kernel<GS, VEC, M, true>), grid, dim3(EMB_SLOT_THREADS), 0, s, a); \
                           else hipLaunchKernelGGL((emb_slot_kernel<GS, VEC, M, false>), grid, dim3(EMB_SLOT_THREADS), 0, s, a); } while (0)
  if (mode == 0) GOCTR_SLOT(0); else GOCTR_SLOT(1);
#undef GOCTR_SLOT
  GOCTR_HIP(hipGetLastError());
  if (direct) {
    const long long borders = std::max<long long>(cdiv(a.B * (long long)(a.T + 1), wgp), 1);      // (upper bound over the batches)
    hipLaunchKernelGGL(emb_span_apply_kernel, dim3((unsigned)cdiv(borders * a.D, 256)), dim3(256), 0, s, a, wgp);
    GOCTR_HIP(hipGetLastError());
  }
  return 0;
}
// layout of the slot kernel: four components per lane (16-byte loads) when the widths allow, else one
// (measured, GOCTR_EMB_SLOT_VEC=4 / 1 forces either: mean pooling at cfg4 80.7 -> 78.3 us with four components per lane; DIN at
// cfg3 got SLOWER, 30.7 -> 33.5 us -- fewer, fatter wavefronts hide less of the latency that bounds it -- so DIN keeps one)
bool emb_slot_vec4(const goctr_model* m) {
  const goctr_ctr_cfg& c = m->cfg;
  const int want = c.kind != GOCTR_DIN ? 4 : 1;
  return (c.D == 16 || c.D == 32 || c.D == 64) && want == 4;
}

int launch_emb_exchange(goctr_model* m, const EmbTrainArgs& a);


// First half of the plan path, in attn_bwd's place in the backward: dpv = dz0 . W0[U:U+2D,:]^T and (DIN) the per-pair
// coefficients -- the kernel gathers every behaviour row and forms dp . x_t like attn_bwd_kernel, so it writes attn_bwd's
// output (the per-sample terms of the att0 gradient, consumed by the weight-gradient launch) as well: one launch instead of two.
int launch_emb_plan_early(goctr_model* m, const RowSource& src, int B, const StepState* st) {
  const goctr_ctr_cfg& c = m->cfg;
  hipStream_t s = engine().stream;
  const int Np = round_up(2 * c.D, 16);
  if (!m->dpv_from_chain) {
    EpiStore sp{m->dpv.p, Np};
    if (launch_nn(GOCTR_K_EMB_TRAIN, m->dz0.p, m->H1p, m->W0pvT.p, Np, B, m->H1p, Np, sp)) return -1;
  }
  if (c.kind != GOCTR_DIN) return 0;
  const int mode = c.att == GOCTR_ATT_COSINE ? 1 : 2;
  ProfScope ps(GOCTR_K_ATTN_BWD);
  if (ps.on) { static char sym[40]; snprintf(sym, sizeof sym, "emb_coef_kernel<%d,%d>", c.D / 4, mode); prof_note_kernel(GOCTR_K_ATTN_BWD, sym); }
  EmbCoefArgs ca{src, st, B, c.T, c.D, m->dpv.p, Np, m->gate_p(m->stp), m->W.p + m->offa, m->emb_dx.p, m->emb_gsum.p,
                 m->wgt_p(m->stp), m->attp.p, m->Tp};
  const dim3 g((unsigned)cdiv(B, 4));
  const int lpr = c.D / 4;
#define GOCTR_COEF(L) do { if (mode == 1) hipLaunchKernelGGL((emb_coef_kernel<L, 1>), g, dim3(256), 0, s, ca); \
                           else hipLaunchKernelGGL((emb_coef_kernel<L, 2>), g, dim3(256), 0, s, ca); } while (0)
  if (lpr == 1) GOCTR_COEF(1); else if (lpr == 2) GOCTR_COEF(2); else if (lpr == 4) GOCTR_COEF(4); else if (lpr == 8) GOCTR_COEF(8); else GOCTR_COEF(16);
#undef GOCTR_COEF
  GOCTR_HIP(hipGetLastError());
  return 0;
}

// Second half (where the table may be written: after every reader of this step): the id-major accumulation over the plan
int launch_emb_plan_step(goctr_model* m, const RowSource& src, int B, const StepState* st, int Np) {
  const goctr_ctr_cfg& c = m->cfg;
  Engine& e = engine();
  hipStream_t s = e.stream;
  const int mode = c.kind != GOCTR_DIN ? 0 : (c.att == GOCTR_ATT_COSINE ? 1 : 2);
  const bool direct = !e.comm_active();
  EmbSlotArgs a{};
  a.plan = m->plan.view(); a.st = st; a.B = B; a.T = c.T; a.D = c.D; a.dpv = m->dpv.p; a.ldp = Np;
  a.dx = m->emb_dx.p; a.gsum = m->emb_gsum.p;
  a.emb = const_cast<float*>(src.emb); a.accum = m->emb_accum.p; a.lr = m->emb_lr;
  {
    ProfScope ps(GOCTR_K_EMB_GRAD);
    if (ps.on) {
      static char sym[48];
      const bool v4 = emb_slot_vec4(m);
      snprintf(sym, sizeof sym, "emb_slot_kernel<%d,%d,%d,%s>", v4 ? c.D / 4 : (c.D <= 16 ? 16 : c.D <= 32 ? 32 : 64), v4 ? 4 : 1, mode ? 1 : 0,
               direct ? "true" : "false");
      prof_note_kernel(GOCTR_K_EMB_GRAD, sym);
    }
    const long long mp = m->plan.max_pairs;
    int rc;
    if (emb_slot_vec4(m)) rc = c.D == 16 ? launch_emb_slot_gv<4, 4>(mode, direct, mp, s, a) : c.D == 32 ? launch_emb_slot_gv<8, 4>(mode, direct, mp, s, a)
                                                                                                        : launch_emb_slot_gv<16, 4>(mode, direct, mp, s, a);
    else rc = c.D <= 16 ? launch_emb_slot_gv<16, 1>(mode, direct, mp, s, a) : c.D <= 32 ? launch_emb_slot_gv<32, 1>(mode, direct, mp, s, a)
                                                                                         : launch_emb_slot_gv<64, 1>(mode, direct, mp, s, a);
    if (rc) return -1;
  }
  if (direct) return 0;
  const int cus = e.compute_units > 0 ? e.compute_units : 256;
  if (m->ex_fixed) {
    // fixed-size buckets: pack the send buffers; the collectives and the owner's side follow from the step driver
    // (emb_exchange_* below), with no host read-back anywhere
    const long long n = (long long)e.eff_world() * m->ex_S * c.D;
    hipLaunchKernelGGL(emb_pack_send_kernel, dim3((unsigned)std::min<long long>(std::max<long long>(cdiv(n, 256), 1), 8 * cus)), dim3(256), 0, s,
                       m->plan.view(), st, m->ex_bucket_off.p, e.eff_world(), m->ex_S, c.D, m->emb_accum.p, m->ex_send_ids.p, m->ex_send_rows.p);
    GOCTR_HIP(hipGetLastError());
    return 0;
  }
  // data parallel without fixed bounds: the exchange reads the batch's slot -> id list and count from fixed buffers
  hipLaunchKernelGGL(emb_plan_select_kernel, dim3((unsigned)std::min<long long>(std::max<long long>(cdiv(m->plan.max_slots, 256), 1), 4 * cus)), dim3(256), 0, s,
                     m->plan.view(), st, m->emb_slot_id.p, m->emb_total.p);
  GOCTR_HIP(hipGetLastError());
  EmbTrainArgs ea{};
  ea.D = c.D; ea.lr = m->emb_lr; ea.emb = const_cast<float*>(src.emb);
  return launch_emb_exchange(m, ea);
}

// Sparse embedding update of one step (emb_train.h).  Runs after every reader of the table in this step (attn_fwd,
// attn_bwd's re-gather) and before the step state advances.
int launch_emb_train(goctr_model* m, const RowSource& src, int B, const StepState* st) {
  const goctr_ctr_cfg& c = m->cfg;
  Engine& e = engine();
  GOCTR_CHECK(src.id_mode, "embedding training needs an id-mode dataset (the dense TrainSample rows carry no ids)");
  GOCTR_CHECK(c.D <= 64, "embedding training supports D <= 64 (got %d)", c.D);
  const int Np = round_up(2 * c.D, 16);
  const long long V = src.V;
  const long long cap = std::min<long long>(V, (long long)B * (c.T + 1));
  GOCTR_CHECK(m->emb_V == V && m->emb_B == B && m->emb_world == e.eff_world() && m->emb_comm == e.comm_active(),
              "embedding-training workspace not prepared (ensure_emb_workspace)");
  const int W = e.comm_active() ? e.eff_world() : 1;
  EmbTrainArgs a{};
  a.src = src; a.st = st; a.B = B; a.T = c.T; a.D = c.D; a.kind = c.kind; a.att = c.att;
  a.dpv = m->dpv.p; a.ldp = Np; a.gate = m->gate_p(m->stp); a.wgt = m->wgt_p(m->stp); a.att0 = m->W.p + m->offa;
  a.emb = const_cast<float*>(src.emb); a.V = V;
  a.mark = m->emb_mark.p; a.rank = m->emb_rank.p; a.accum = m->emb_accum.p; a.lr = m->emb_lr; a.dbg = 0;
  a.W = W; a.Vw = m->emb_Vw;
  hipStream_t s = e.stream;
  if (m->plan.valid) {
    // the id-major path: no marks, no scans, no accumulators to apply -- the dpv GEMM, then the plan kernels.
    // (W0[U:U+2D,:]^T is transposed once per call sequence -- ensure_w0pv, outside the captured step -- and then kept
    // current by the Adam kernels like the other operand copies: 4.2 us per step less)
    // (the dpv GEMM and the coefficient kernel already ran in attn_bwd's place: launch_emb_plan_early)
    return launch_emb_plan_step(m, src, B, st, Np);
  }
  {
  ProfScope ps(GOCTR_K_EMB_TRAIN);
  // ids with a single occurrence are applied in place (emb_train.h); only without a communicator (another rank may touch
  // the id too) and when the vocabulary is larger than the batch's id count (otherwise hardly any id is single)
  const long long pairs = (long long)B * (c.T + 1);
  const int singles = env_int("GOCTR_EMB_SINGLES", (!e.comm_active() && V > pairs) ? 1 : 0) != 0 && !e.comm_active();
  hipLaunchKernelGGL(emb_mark_kernel, dim3((unsigned)cdiv(pairs, 256)), dim3(256), 0, s, a, singles);
  if (singles) hipLaunchKernelGGL(emb_mark2_kernel, dim3((unsigned)cdiv(pairs, 256)), dim3(256), 0, s, a);
  GOCTR_HIP(hipGetLastError());
  // rank scan over the owner-major index space: this rank's touched ids get dense slots, bucket after bucket (owner =
  // id % world), ascending ids inside a bucket
  if (exclusive_scan_sink(m->emb_mark.p, m->emb_Vw * W, m->emb_tiles, m->emb_total.p, EmbMultiMap{},
                          EmbRankSink{m->emb_mark.p, m->emb_rank.p, m->emb_slot_id.p, W, m->emb_Vw, 0})) return -1;
  hipLaunchKernelGGL(w0pv_transpose_kernel, dim3((unsigned)cdiv((long long)m->H1p * Np, 256)), dim3(256), 0, s, m->W.p, m->H1p,
                     c.U, 2 * c.D, Np, m->W0pvT.p);
  GOCTR_HIP(hipGetLastError());
  }
  EpiStore sp{m->dpv.p, Np};
  if (launch_nn(GOCTR_K_EMB_TRAIN, m->dz0.p, m->H1p, m->W0pvT.p, Np, B, m->H1p, Np, sp)) return -1;
  // attention modes: one 1024-thread workgroup per CU (~90 VGPRs allow no second one) with a <= 136 KB LDS cache of hot
  // rows; mean pooling fits two per CU (GOCTR_EMB_WGS=2, <= 72 KB each) but measured no faster (184 vs 178 us at cfg4)
  const int mode = c.kind != GOCTR_DIN ? 0 : (c.att == GOCTR_ATT_COSINE ? 1 : 2);
  const int wg_per_cu = 1;
  const size_t budget = wg_per_cu > 1 ? 72u * 1024u : 136u * 1024u;
  int nslot = 1;
  while ((size_t)nslot * 2 * (c.D * sizeof(long long) + sizeof(int)) <= budget) nslot *= 2;
  const size_t lds = (size_t)nslot * (c.D * sizeof(long long) + sizeof(int));
  const int cus = e.compute_units > 0 ? e.compute_units : 256;
  const dim3 gb((unsigned)std::min<long long>(cdiv(B, EMB_GRAD_THREADS / 64), (long long)wg_per_cu * cus));
  // GOCTR_EMB_CACHE=0 (experiments) sends every add straight to HBM: 5x slower at cfg3 AND at cfg4 — a Zipfian head is
  // hot in a 10^7-row vocabulary too
  const bool cache = true;
  const dim3 gg = cache ? gb : dim3((unsigned)std::min<long long>(cdiv(B, EMB_GRAD_THREADS / 64), 8 * cus));
  {
    ProfScope ps(GOCTR_K_EMB_GRAD);
    if (ps.on) {
      static char sym[48];
      snprintf(sym, sizeof sym, "emb_grad_kernel<%d,%d,%s>", c.D <= 16 ? 16 : c.D <= 32 ? 32 : 64, mode, cache ? "true" : "false");
      prof_note_kernel(GOCTR_K_EMB_GRAD, sym);
    }
    if (c.D <= 16) launch_emb_grad<16>(mode, cache, gg, lds, s, a, nslot);
    else if (c.D <= 32) launch_emb_grad<32>(mode, cache, gg, lds, s, a, nslot);
    else launch_emb_grad<64>(mode, cache, gg, lds, s, a, nslot);
  }
  if (e.comm_active()) return launch_emb_exchange(m, a);
  ProfScope ps(GOCTR_K_EMB_TRAIN);
  hipLaunchKernelGGL(emb_apply_kernel, dim3((unsigned)std::min<long long>(cdiv(cap * c.D, 256), 16 * cus)), dim3(256), 0, s, a,
                     m->emb_slot_id.p, m->emb_total.p);
  GOCTR_HIP(hipGetLastError());
  return 0;
}

// backward part up to and including the slab reduce (optionally fused with Adam on a single GPU)
// ---- the fixed-size exchange of a data-parallel step with trainable embeddings, piece by piece (emb_train.h, end):
//   [graph 1: forward, backward, plan kernels, emb_pack_send]  ->  emb_exchange_a2a  ->  [graph 2: emb_exchange_owner, slab
//   reduce]  ->  emb_exchange_gather + the dense all-reduce  ->  [graph 3: emb_exchange_apply, Adam]
bool emb_split3(const goctr_model* m) { return engine().comm_active() && m->emb_lr > 0.f && m->plan.valid && m->ex_fixed; }
// uniform all-to-all: S (id, row) entries to and from every rank
int emb_exchange_a2a(goctr_model* m) {
  Engine& e = engine();
  const int W = e.eff_world(), D = m->cfg.D;
  std::vector<size_t> off((size_t)W), cnt((size_t)W), offD((size_t)W), cntD((size_t)W);
  for (int p = 0; p < W; ++p) { off[p] = (size_t)p * m->ex_S; cnt[p] = (size_t)m->ex_S; offD[p] = off[p] * D; cntD[p] = cnt[p] * D; }
  ProfScope ps(GOCTR_K_ALLREDUCE);
  if (comm_alltoallv(m->ex_send_ids.p, off.data(), cnt.data(), m->ex_recv_ids.p, off.data(), cnt.data(), 4)) return -1;
  return comm_alltoallv(m->ex_send_rows.p, offD.data(), cntD.data(), m->ex_recv_rows.p, offD.data(), cntD.data(), 8);
}
// owner: unique ids of my bucket among the W * S received entries -> dense slots, exact integer sums, deltas, padded id list
int emb_exchange_owner(goctr_model* m) {
  Engine& e = engine();
  const int W = e.eff_world(), r = e.eff_rank(), D = m->cfg.D;
  hipStream_t s = e.stream;
  const long long nrecv = (long long)W * m->ex_S;
  const int cus = e.compute_units > 0 ? e.compute_units : 256;
  ProfScope ps(GOCTR_K_EMB_TRAIN);
  hipLaunchKernelGGL(emb_recv_mark_kernel, dim3((unsigned)cdiv(nrecv, 256)), dim3(256), 0, s, m->ex_recv_ids.p, nrecv, W, m->emb_Vw, m->emb_mark.p);
  GOCTR_HIP(hipGetLastError());
  if (exclusive_scan_sink(m->emb_mark.p + (size_t)r * m->emb_Vw, m->emb_Vw, m->emb_tiles, m->ex_red_total.p, EmbMultiMap{},
                          EmbRankSink{m->emb_mark.p, m->emb_rank.p, m->ex_red_ids.p, W, m->emb_Vw, (long long)r * m->emb_Vw})) return -1;
  hipLaunchKernelGGL(emb_recv_accumulate_kernel, dim3((unsigned)cdiv(nrecv * D, 256)), dim3(256), 0, s, m->ex_recv_ids.p, m->ex_recv_rows.p,
                     nrecv, D, W, m->emb_Vw, m->emb_rank.p, m->ex_red.p);
  hipLaunchKernelGGL(emb_delta_kernel, dim3((unsigned)std::min<long long>(std::max<long long>(cdiv((long long)m->ex_R * D, 256), 1), 16 * cus)), dim3(256), 0, s,
                     m->ex_red.p, m->ex_red_total.p, D, m->emb_lr, m->ex_delta.p);
  hipLaunchKernelGGL(emb_pad_ids_kernel, dim3((unsigned)std::min<long long>(std::max<long long>(cdiv((long long)m->ex_R, 256), 1), 4 * cus)), dim3(256), 0, s,
                     m->ex_red_ids.p, m->ex_red_total.p, m->ex_R);
  GOCTR_HIP(hipGetLastError());
  return 0;
}
// every owner's R (id, delta) entries to every rank
int emb_exchange_gather(goctr_model* m) {
  const int D = m->cfg.D;
  ProfScope ps(GOCTR_K_ALLREDUCE);
  if (comm_allgather_i32(m->ex_red_ids.p, m->ex_gids.p, (size_t)m->ex_R)) return -1;
  return comm_allgather_i32(reinterpret_cast<const int*>(m->ex_delta.p), reinterpret_cast<int*>(m->ex_gdelta.p), (size_t)m->ex_R * D);
}
// every replica applies every delta (ids are unique across the owners' lists; -1 = padding)
int emb_exchange_apply(goctr_model* m, const RowSource& src) {
  Engine& e = engine();
  const int D = m->cfg.D;
  const long long ng = (long long)e.eff_world() * m->ex_R;
  const int cus = e.compute_units > 0 ? e.compute_units : 256;
  ProfScope ps(GOCTR_K_EMB_TRAIN);
  hipLaunchKernelGGL(emb_apply_gathered_kernel, dim3((unsigned)std::min<long long>(std::max<long long>(cdiv(ng * D, 256), 1), 16 * cus)), dim3(256), 0, e.stream,
                     const_cast<float*>(src.emb), m->ex_gids.p, m->ex_gdelta.p, ng, D);
  GOCTR_HIP(hipGetLastError());
  return 0;
}

int launch_reduce_part(goctr_model* m, const RowSource& src, int B, const StepOpts& o, bool advance, bool fuse_update, const ReduceArgs& ra);
// stage 0: the whole backward; 1: everything before the slab reduce (the sparse embedding update ends with its send buffers
// packed); 2: the slab reduce alone -- the two halves of a data-parallel step with trainable embeddings, whose all-to-all
// runs between them (split3 below)
int launch_backward(goctr_model* m, const RowSource& src, int B, const StepOpts& o, bool advance,
                    bool fuse_update = false, int stage = 0) {
  const goctr_ctr_cfg& c = m->cfg;
  Engine& e = engine();
  if (stage == 2) return launch_reduce_part(m, src, B, o, advance, fuse_update, m->pend_ra);
  const StepState* st = m->st_cur();
  const uint32_t row_off = (uint32_t)(e.eff_rank() * B);
  const bool drop = o.drop_mode != 0;
  DropCfg d0{drop && o.p0 > 0 ? o.drop_mode : 0, o.p0, m->mask0.p, c.H1, o.seed, 0u, row_off};
  DropCfg d1{drop && o.p1 > 0 ? o.drop_mode : 0, o.p1, m->mask1.p, c.H2, o.seed, 1u, row_off};
  const float* A0 = d0.mode ? m->A0.p : m->P0.p;
  const float* A1 = d1.mode ? m->A1.p : m->P1.p;
  const TnSchedule ts = tn_schedule(m, B);
  const int rpw = ts.rows, S = ts.S, rpl = ts.rows_light, SL = ts.S_light;

  const bool fused = chain_ok(m);  // dz1 / dz0 / dp were already produced by the chain kernel
  if (fused) {
    A0 = m->A0.p; A1 = m->A1.p;    // the chain kernel always writes the post-dropout activations here
  } else {
    EpiDSig b1{m->dz1.p, m->P1.p, m->H2p, c.H2, d1, st};
    if (launch_nn(GOCTR_K_BWD_DZ1, m->dz2.p, 16, m->W2T.p, m->H2p, B, 16, m->H2p, b1)) return -1;
    EpiDSig b0{m->dz0.p, m->P0.p, m->H1p, c.H1, d0, st};
    if (launch_nn(GOCTR_K_BWD_DZ0, m->dz1.p, m->H2p, m->W1T.p, m->H1p, B, m->H2p, m->H1p, b0)) return -1;
  }
  if (c.kind == GOCTR_DIN) {
    if (!fused) {
      EpiStore sp{m->dp.p, m->Dp};
      if (launch_nn(GOCTR_K_BWD_DP, m->dz0.p, m->H1p, m->W0sT.p, m->Dp, B, m->H1p, m->Dp, sp)) return -1;
    }
    if (emb_plan_active(m) && src.id_mode) {
      if (launch_emb_plan_early(m, src, B, st)) return -1;      // (does attn_bwd's job too)
    } else if (fused && m->attn_bwd_in_chain) {
      // (the chain kernel's tail wrote the terms, launch_chain_x3)
    } else {
      AttnBwdArgs ab{};
      ab.src = src; ab.st = st; ab.B = B; ab.T = c.T; ab.D = c.D; ab.Dp = m->Dp; ab.Tp = m->Tp;
      ab.dp = m->dp.p; ab.gate = m->gate_p(m->stp); ab.wgt = m->wgt_p(m->stp); ab.partial = m->attp.p;
      if (launch_attn_bwd(ab, (int)cdiv(B, ATTN_BWD_WAVES))) return -1;
    }
  } else if (emb_plan_active(m) && src.id_mode) {
    if (launch_emb_plan_early(m, src, B, st)) return -1;
  }

  // weight gradients: all GEMMs in one launch; dW1 and dW2 are posed transposed, datt0 is a ones-column
  // product over the per-sample terms (see mfma_gemm.h: gemm_tn_multi_kernel)
  int nt_max = m->H1p / 16;
  if (m->H2p / 16 > nt_max) nt_max = m->H2p / 16;
  if (c.kind == GOCTR_DIN && m->Tp / 16 > nt_max) nt_max = m->Tp / 16;
  const bool multi = nt_max <= 16 && gemm_tn_multi_fits<3, GOCTR_TN_CH>(nt_max);
  const TnWide tw = (multi && GOCTR_TN_CH == 32)
                        ? tn_schedule_wide(m, B, (m->dw2_from_chain ? 1 : 0) + (m->att0_from_chain ? 1 : 0)) : TnWide{};
  int S0 = S, S1 = S, SLx = SL;      // slabs per segment, for the reduce below
  int SL2x = -1, SL3x = -1;          // (wide launch: the dW2 / att0 segments' own slab counts)
  if (multi && tw.ok) {
    TnMulti tm{};
    tm.M = B; tm.np = 3; tm.wt = env_int("GOCTR_TN_WT", 2);
    const int b0 = tw.kblocks0 * 2 * tw.S0, b1 = 2 * tw.S1;
    tm.p[0] = {m->h0.p, m->Ip, m->Ip / 16, m->dz0.p, m->H1p, m->H1p / 16, m->slabs0.p,
               (unsigned long long)m->Ip * m->H1p, 0, m->H1p, 0, tw.rows0, tw.S0, 2, tw.nbt};
    tm.p[1] = {m->dz1.p, m->H2p, m->H2p / 16, A0, m->H1p, m->H1p / 16, m->slabs1.p,
               (unsigned long long)m->H1p * m->H2p, 1, m->H2p, b0, tw.rows1, tw.S1, 2, tw.nbt};
    // (round 6) where the chain launch left per-tile sums the problem is a SUM problem: KT = 0, D = the partials [tiles][lda],
    // `rows` = tiles per slab, TN_SUM_SLABS slabs
    const int ntiles = (int)cdiv(B, 32);
    const int nsum = std::min(TN_SUM_SLABS, ntiles), tps = (int)cdiv(ntiles, nsum);
    int SL2 = tw.SL, SL3 = tw.SL;
    if (m->dw2_from_chain) {
      SL2 = (int)cdiv(ntiles, tps);
      tm.p[2] = {nullptr, m->H2p, 0, m->tile_dw2.p, m->H2p, 0, m->slabs2.p, (unsigned long long)m->H2p * 16, 1, 16, b0 + b1, tps, SL2, 0, 0};
    } else {
      tm.p[2] = {m->dz2.p, 16, 1, A1, m->H2p, m->H2p / 16, m->slabs2.p, (unsigned long long)m->H2p * 16, 1, 16,
                 b0 + b1, tw.rowsL, tw.SL, 0, 0};
    }
    int nblk = b0 + b1 + SL2;
    // att0's update inside this launch (ctr_chain_x3.h att0_early_body): the single-GPU pipelined step with the per-tile sums of the
    // att0 terms, where the step's last launch also runs the next batch's attention -- which then needs no flag (launch_reduce_part).
    // GOCTR_ATT0_EARLY=0: the sum problem below + the reduce block + the flag, as until round 6's last session.
    m->att0_early = c.kind == GOCTR_DIN && m->att0_from_chain && o.pipelined && fuse_update && stage == 0 && !e.comm_active() &&
                    m->Tp % 32 == 0 && env_int("GOCTR_ATT0_EARLY", 1) != 0;
    Att0EarlyArgs ae{};
    if (m->att0_early) {
      ae.tile_att0 = m->tile_att0.p; ae.ntiles = ntiles; ae.Tp = m->Tp; ae.tps = tps; ae.nslabs = (int)cdiv(ntiles, tps);
      ae.ad = make_adam_args(m, B, *o.tc); ae.st = st;
    } else
    if (c.kind == GOCTR_DIN) {  // datt0 = ones^T . dgs  (column sums over the batch)
      if (m->att0_from_chain) {
        SL3 = (int)cdiv(ntiles, tps);
        tm.p[3] = {nullptr, m->Tp, 0, m->tile_att0.p, m->Tp, 0, m->slabs3.p, (unsigned long long)16 * m->Tp, 0, m->Tp, nblk, tps, SL3, 0, 0};
      } else {
        tm.p[3] = {m->ones16.p, 16, 1, m->attp.p, m->Tp, m->Tp / 16, m->slabs3.p, (unsigned long long)16 * m->Tp, 0, m->Tp,
                   nblk, tw.rowsL, tw.SL, 0, 0};
      }
      tm.np = 4;
      nblk += SL3;
    }
    static DevBuf<unsigned long long> tndbgw;
    const bool dbg = dbg_on("tn");
    if (dbg && !tndbgw.p && tndbgw.alloc(16)) return -1;
    tm.dbg = dbg ? tndbgw.p : nullptr;
    {
      ProfScope ps(GOCTR_K_DW0);
      if (ps.on) prof_note_kernel(GOCTR_K_DW0, tw.ktw0 == 9 ? "gemm_tn_multi_x3w_kernel<9,5>" : "gemm_tn_multi_x3w_kernel<8,5>");
      if (m->att0_early) {      // (one more workgroup, the last: att0's sum and update)
        if (ps.on) prof_note_kernel(GOCTR_K_DW0, tw.ktw0 == 9 ? "gemm_tn_multi_x3w_att0_kernel<9,5>" : "gemm_tn_multi_x3w_att0_kernel<8,5>");
        if (tw.ktw0 == 9)
          hipLaunchKernelGGL((gemm_tn_multi_x3w_att0_kernel<9, 5>), dim3((unsigned)nblk + 1), dim3(512), gemm_tn_multi_x3w_lds_bytes<9>(), e.stream, tm, ae);
        else
          hipLaunchKernelGGL((gemm_tn_multi_x3w_att0_kernel<8, 5>), dim3((unsigned)nblk + 1), dim3(512), gemm_tn_multi_x3w_lds_bytes<8>(), e.stream, tm, ae);
      } else
      if (tw.ktw0 == 9)
        hipLaunchKernelGGL((gemm_tn_multi_x3w_kernel<9, 5>), dim3((unsigned)nblk), dim3(512), gemm_tn_multi_x3w_lds_bytes<9>(), e.stream, tm);
      else
        hipLaunchKernelGGL((gemm_tn_multi_x3w_kernel<8, 5>), dim3((unsigned)nblk), dim3(512), gemm_tn_multi_x3w_lds_bytes<8>(), e.stream, tm);
      GOCTR_HIP(hipGetLastError());
    }
    if (dbg) {
      unsigned long long h[16];
      if (tndbgw.download(h, 16)) return -1;
      fprintf(stderr, "dW x3w (rows %d/%d/%d, %d workgroups) wg 0: multiplier wave: wait for chunk 0 %lld, in MFMA sections %lld, loop total %lld, "
              "epilogue %lld | stager wave: first chunk %lld, staging sections %lld, total %lld, chunks %lld (s_memtime ticks)\n",
              tw.rows0, tw.rows1, tw.rowsL, nblk, (long long)h[0], (long long)h[1], (long long)h[2], (long long)h[3], (long long)h[8],
              (long long)h[9], (long long)h[10], (long long)h[11]);
    }
    S0 = tw.S0; S1 = tw.S1; SLx = tw.SL; SL2x = SL2; SL3x = SL3;
  } else if (multi) {
    TnMulti tm{};
    tm.M = B; tm.np = 3;
    const int kb0 = (int)cdiv(m->Ip / 16, 3), kb1 = (int)cdiv(m->H2p / 16, 3), kb2 = 1;
    tm.p[0] = {m->h0.p, m->Ip, m->Ip / 16, m->dz0.p, m->H1p, m->H1p / 16, m->slabs0.p,
               (unsigned long long)m->Ip * m->H1p, 0, m->H1p, 0, rpw, S};
    tm.p[1] = {m->dz1.p, m->H2p, m->H2p / 16, A0, m->H1p, m->H1p / 16, m->slabs1.p,
               (unsigned long long)m->H1p * m->H2p, 1, m->H2p, kb0 * S, rpw, S};
    tm.p[2] = {m->dz2.p, 16, 1, A1, m->H2p, m->H2p / 16, m->slabs2.p, (unsigned long long)m->H2p * 16, 1, 16,
               (kb0 + kb1) * S, rpl, SL};
    int nblk = (kb0 + kb1) * S + kb2 * SL;
    if (c.kind == GOCTR_DIN) {  // datt0 = ones^T . dgs  (column sums over the batch)
      tm.p[3] = {m->ones16.p, 16, 1, m->attp.p, m->Tp, m->Tp / 16, m->slabs3.p, (unsigned long long)16 * m->Tp, 0, m->Tp,
                 nblk, rpl, SL};
      tm.np = 4;
      nblk += SL;
    }
    const size_t lds_tm = gemm_tn_multi_lds_bytes<3, 4, GOCTR_TN_CH>(nt_max);
    static DevBuf<unsigned long long> tndbg;
    const bool dbg = dbg_on("tn");
    if (dbg && !tndbg.p && tndbg.alloc(16)) return -1;
    tm.dbg = dbg ? tndbg.p : nullptr;
    {
      ProfScope ps(GOCTR_K_DW0);
      if (ps.on) prof_note_kernel(GOCTR_K_DW0, (GOCTR_TN_CH == 32) ? "gemm_tn_multi_x3_kernel<3,4>" : "gemm_tn_multi_kernel<3,4,32>");
      // default: the 6-product bf16 split (mfma_gemm.h); GOCTR_TN_F32=1 selects the v_mfma_f32_16x16x4_f32 body
      if (GOCTR_TN_CH == 32)
        hipLaunchKernelGGL((gemm_tn_multi_x3_kernel<3, 4>), dim3((unsigned)nblk), dim3(512), gemm_tn_multi_x3_lds_bytes<3>(nt_max), e.stream, tm);
      else
        hipLaunchKernelGGL((gemm_tn_multi_kernel<3, 4, GOCTR_TN_CH>), dim3((unsigned)nblk), dim3(256), lds_tm, e.stream, tm);
      GOCTR_HIP(hipGetLastError());
    }
    if (dbg) {
      unsigned long long h[16];
      if (tndbg.download(h, 16)) return -1;
      fprintf(stderr, "dW x3 wg 0: multiplier wave: wait for chunk 0 %lld, in MFMA sections %lld, loop total %lld, epilogue %lld | stager wave: first chunk %lld, "
              "staging sections %lld, total %lld, chunks %lld (s_memtime ticks)\n", (long long)h[0], (long long)h[1], (long long)h[2], (long long)h[3],
              (long long)h[8], (long long)h[9], (long long)h[10], (long long)h[11]);
    } else if (dbg) {
      unsigned long long h[16];
      if (tndbg.download(h, 16)) return -1;
      for (int w = 0; w < 2; ++w)
        fprintf(stderr, "dW wg %s: rows %d S %d nblk %d | prologue %lld, first chunk mma %lld, loop %lld, epilogue %lld, total %lld\n",
                w ? "last" : "0", rpw, S, nblk, (long long)(h[8 * w + 1] - h[8 * w]), (long long)(h[8 * w + 2] - h[8 * w + 1]),
                (long long)(h[8 * w + 3] - h[8 * w + 1]), (long long)(h[8 * w + 4] - h[8 * w + 3]), (long long)(h[8 * w + 4] - h[8 * w]));
    }
  } else {
    if (launch_tn(GOCTR_K_DW0, m->h0.p, m->Ip, m->Ip / 16, m->dz0.p, m->H1p, m->H1p / 16, B, rpw, m->slabs0.p,
                  (size_t)m->Ip * m->H1p)) return -1;
    if (launch_tn(GOCTR_K_DW1, A0, m->H1p, m->H1p / 16, m->dz1.p, m->H2p, m->H2p / 16, B, rpw, m->slabs1.p,
                  (size_t)m->H1p * m->H2p)) return -1;
    if (launch_tn(GOCTR_K_DW2, A1, m->H2p, m->H2p / 16, m->dz2.p, 16, 1, B, rpw, m->slabs2.p, (size_t)m->H2p * 16)) return -1;
    if (c.kind == GOCTR_DIN &&
        launch_tn(GOCTR_K_DW2, m->ones16.p, 16, 1, m->attp.p, m->Tp, m->Tp / 16, B, rpw, m->slabs3.p, (size_t)16 * m->Tp))
      return -1;
  }

  if (m->emb_lr > 0.f && launch_emb_train(m, src, B, st)) return -1;

  ReduceArgs ra{};
  ra.seg[0] = {m->slabs0.p, S0, (unsigned long long)m->Ip * m->H1p, 0, m->Ip * m->H1p};
  ra.seg[1] = {m->slabs1.p, S1, (unsigned long long)m->H1p * m->H2p, m->off1, m->H1p * m->H2p};
  ra.seg[2] = {m->slabs2.p, SL2x > 0 ? SL2x : (multi ? SLx : S), (unsigned long long)m->H2p * 16, m->off2, m->H2p * 16};
  ra.nseg = 3;
  if (c.kind == GOCTR_DIN && !m->att0_early) {      // (att0_early: no slabs, the weight-gradient launch has updated att0 itself)
    ra.seg[3] = {m->slabs3.p, SL3x > 0 ? SL3x : (multi ? SLx : S), (unsigned long long)16 * m->Tp, m->offa, m->Tp};
    ra.nseg = 4;
  }
  ra.nflat = m->nflat; ra.lossrow = m->lossrow.p; ra.B = B; ra.G = m->G.p; ra.st = m->st_cur(); ra.st_out = m->st_next(); ra.advance = advance ? 1 : 0;
  if (stage == 1) { m->pend_ra = ra; return 0; }
  return launch_reduce_part(m, src, B, o, advance, fuse_update, ra);
}

int launch_reduce_part(goctr_model* m, const RowSource& src, int B, const StepOpts& o, bool advance, bool fuse_update, const ReduceArgs& ra) {
  const goctr_ctr_cfg& c = m->cfg;
  Engine& e = engine();
  if (fuse_update) {
    ReduceAdamArgs p{};
    p.r = ra; p.ad = make_adam_args(m, B, *o.tc);
    p.ra_flag = m->ra_flag.p; p.ra_block = (o.pipelined && c.kind == GOCTR_DIN) ? (m->offa * 2) / 256 : -1;
    if (m->att0_early) {      // att0 is already this step's: no flag for the attention part, and the reduce part keeps its hands off
      p.ra_flag = nullptr; p.ra_block = -1; p.skip_begin = m->offa; p.skip_len = m->Tp;
    }
    ProfScope ps(GOCTR_K_REDUCE);
    if (o.pipelined) {
      if (launch_reduce_attn(m, src, B, o, p)) return -1;     // + attn_fwd of the next step's batch
    } else {
      hipLaunchKernelGGL(reduce_adam_kernel, dim3((unsigned)cdiv((int64_t)m->nflat * 2, 256) + 1), dim3(256), 0, e.stream, p);
      GOCTR_HIP(hipGetLastError());
    }
    m->stp ^= 1;   // the step is closed: later launches read the slot just written
    return 0;
  }
  {
    ProfScope ps(GOCTR_K_REDUCE);
    hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)cdiv((int64_t)m->nflat * 2, 256) + 1), dim3(256), 0, e.stream, ra);
    GOCTR_HIP(hipGetLastError());
  }
  if (advance) m->stp ^= 1;
  return 0;
}

AdamArgs make_adam_args(goctr_model* m, int B, const goctr_train_cfg& tc) {
  Engine& e = engine();
  AdamArgs a{};
  a.W = m->W.p; a.G = m->G.p; a.Mo = m->Mo.p; a.Vo = m->Vo.p; a.nflat = m->nflat;
  a.off1 = m->off1; a.off2 = m->off2; a.offa = m->offa;
  a.Ip = m->Ip; a.H1p = m->H1p; a.H2p = m->H2p; a.Dp = m->Dp; a.U = m->cfg.U; a.D = m->cfg.D;
  a.W1T = m->W1T.p; a.W2T = m->W2T.p; a.W0sT = m->W0sT.p;
  a.W0i = m->img(0); a.W1i = m->img(1); a.W1Ti = m->img(2); a.W0sTi = m->img(3);
  a.x3 = m->x3_images();
  // (kept in step with W0 once embedding training has built it: launch_emb_train transposes it once, Adam keeps it current)
  a.W0pvT = (m->emb_lr > 0.f && m->w0pv_live) ? m->W0pvT.p : nullptr; a.Npv = round_up(2 * m->cfg.D, 16);
  a.lr = tc.lr; a.l2 = tc.l2; a.beta1 = tc.beta1; a.beta2 = tc.beta2; a.eps = tc.eps;
  a.div_by_batch = tc.adam_div_by_batch; a.l2_first = tc.adam_l2_before_batch_div;
  a.bglobal = B * e.eff_world(); a.st = m->st_cur(); a.costs = m->costs.p;
  return a;
}

int launch_adam(goctr_model* m, int B, const goctr_train_cfg& tc) {
  AdamArgs a = make_adam_args(m, B, tc);
  ProfScope ps(GOCTR_K_ADAM);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)cdiv(m->nflat, 256) + 1), dim3(256), 0, engine().stream, a);   // (+ the block that prepares the next step's bias corrections)
  GOCTR_HIP(hipGetLastError());
  return 0;
}

// the Adam launch of a step whose reduce and update are separate launches (data parallel: the all-reduce sits between them);
// pipelined: merged with the NEXT step's attention (adam_attn_kernel) -- state and parity are already the new step's
int launch_adam_step(goctr_model* m, const RowSource& src, int B, const StepOpts& o) {
  if (!(o.pipelined && engine().comm_active())) return launch_adam(m, B, *o.tc);
  int groups = 0;
  const int fast = attn_fast_mode(m, src, &groups);
  const AdamArgs ad = make_adam_args(m, B, *o.tc);
  const AttnArgs aa = make_attn_args(m, src, B, m->st_cur(), m->stp, gate_fac_mode(m, src, o, B));
  const int nadam = (int)cdiv(m->nflat, 256) + 1;
  const int ra_block = m->cfg.kind == GOCTR_DIN ? m->offa / 256 : -1;
  const dim3 grid((unsigned)(nadam + cdiv(B, 4))), blk(256);
  hipStream_t st = engine().active;
  ProfScope ps(GOCTR_K_ADAM);
#define GOCTR_AA(L)                                                                                                       \
  do {                                                                                                                    \
    if (fast == 1) hipLaunchKernelGGL((adam_attn_kernel<4, L, 1>), grid, blk, 0, st, ad, m->ra_flag.p, ra_block, aa, nadam);       \
    else if (fast == 2) hipLaunchKernelGGL((adam_attn_kernel<4, L, 2>), grid, blk, 0, st, ad, m->ra_flag.p, ra_block, aa, nadam);  \
    else hipLaunchKernelGGL((adam_attn_kernel<4, L, 3>), grid, blk, 0, st, ad, m->ra_flag.p, ra_block, aa, nadam);                 \
  } while (0)
  if (groups == 4) GOCTR_AA(4);
  else GOCTR_AA(16);
#undef GOCTR_AA
  GOCTR_HIP(hipGetLastError());
  return 0;
}

int allreduce_grads(goctr_model* m) {
  if (!engine().comm_active()) return 0;
  ProfScope ps(GOCTR_K_ALLREDUCE);
  return comm_allreduce_f32(m->G.p, (size_t)m->nflat + 1);
}

// one full training step, eager
int train_step_eager(goctr_model* m, const RowSource& src, int B, const StepOpts& o) {
  if (launch_forward(m, src, B, o)) return -1;
  const bool fuse = !engine().comm_active();
  if (emb_split3(m)) {
    if (launch_backward(m, src, B, o, true, false, 1) || emb_exchange_a2a(m) || emb_exchange_owner(m) ||
        launch_backward(m, src, B, o, true, false, 2) || emb_exchange_gather(m) || allreduce_grads(m) ||
        emb_exchange_apply(m, src)) return -1;
    return launch_adam(m, B, *o.tc);
  }
  if (launch_backward(m, src, B, o, true, fuse)) return -1;
  if (fuse) return 0;
  if (allreduce_grads(m)) return -1;
  return launch_adam(m, B, *o.tc);
}

bool graph_matches(const StepGraph& g, const goctr_dataset* d, const goctr_emb* e, int B, const StepOpts& o, bool fac) {
  return g.fac == fac && g.a[0] && g.a[1] && g.ds == d->uid && g.emb == (e ? e->uid : 0) && g.B == B && g.mode == o.drop_mode && g.p0 == o.p0 && g.p1 == o.p1 &&
         g.seed == o.seed && g.lr == o.tc->lr && g.l2 == o.tc->l2 && g.b1 == o.tc->beta1 && g.b2 == o.tc->beta2 &&
         g.eps == o.tc->eps && g.flags == o.tc->adam_div_by_batch * 2 + o.tc->adam_l2_before_batch_div &&
         g.world == engine().eff_world() && g.comm == engine().comm_active() && g.pipelined == o.pipelined;
}

int build_graph(goctr_model* m, const goctr_dataset* d, const goctr_emb* emb, const RowSource& src, int B,
                const StepOpts& o) {
  Engine& e = engine();
  m->graph.destroy();
  const bool fuse = !e.comm_active();
  const int stp_now = m->stp;
  struct StpGuard {      // every exit path (the GOCTR_HIP returns included) restores the parity and drops a half-built graph set
    goctr_model* m; int stp; bool ok = false;
    ~StpGuard() { m->stp = stp; if (!ok) m->graph.destroy(); }
  } stp_guard{m, stp_now};
  for (int par = 0; par < 2; ++par) {
    m->stp = par;                      // the captured launches bake this parity's state pointers in
    const bool split3 = emb_split3(m);
    // (capture_graph retakes a capture another thread's runtime calls invalidated; `back` = the parity its body starts from)
    int back = m->stp;
    auto restore = [&] { m->stp = back; };
    if (capture_graph(e.stream, &m->graph.a[par], [&] {
          int rc = launch_forward(m, src, B, o) || launch_backward(m, src, B, o, true, fuse, split3 ? 1 : 0);
          if (!rc && !e.comm_active() && !fuse) rc = launch_adam(m, B, *o.tc);
          return rc;
        }, restore)) return -1;
    if (split3) {
      back = m->stp;
      if (capture_graph(e.stream, &m->graph.mid[par], [&] { return emb_exchange_owner(m) || launch_backward(m, src, B, o, true, false, 2); },
                        restore)) return -1;
    }
    if (e.comm_active()) {
      back = m->stp;                   // (flipped by launch_backward: Adam reads the new slot)
      if (capture_graph(e.stream, &m->graph.b[par], [&] { return (split3 && emb_exchange_apply(m, src)) || launch_adam_step(m, src, B, o); },
                        restore)) return -1;
      if (!split3) {
        // b[par] + the next step's a (parity par ^ 1, where m->stp stands now): launch_backward flips m->stp back to par
        back = m->stp;
        if (capture_graph(e.stream, &m->graph.ba[par], [&] {
              return launch_adam_step(m, src, B, o) || launch_forward(m, src, B, o) || launch_backward(m, src, B, o, true, false, 0);
            }, restore)) return -1;
      }
    }
  }
  m->stp = stp_now;
  StepGraph& sg = m->graph;
  sg.ds = d->uid; sg.emb = emb ? emb->uid : 0; sg.B = B; sg.mode = o.drop_mode; sg.p0 = o.p0; sg.p1 = o.p1; sg.seed = o.seed;
  sg.lr = o.tc->lr; sg.l2 = o.tc->l2; sg.b1 = o.tc->beta1; sg.b2 = o.tc->beta2; sg.eps = o.tc->eps;
  sg.flags = o.tc->adam_div_by_batch * 2 + o.tc->adam_l2_before_batch_div; sg.world = e.eff_world(); sg.comm = e.comm_active();
  sg.pipelined = o.pipelined; sg.fac = gate_fac_mode(m, src, o, B);
  stp_guard.ok = true;
  return 0;
}

// Data parallel (dense all-reduce only): may the multi-step graphs hold the collective itself?  RCCL collectives can be
// captured; whether THIS build of RCCL on THIS box replays them correctly is established once per communicator by
// comm_capture_selftest (comm.hip: captured vs eager all-reduce, the verdict agreed on by all ranks), GOCTR_DP_CAPTURE_COMM=0
// switches the mode off, =2 on without the test.  The loop-back communicator's host barriers can never be captured.
bool dp_capture_ok(const goctr_model* m) {
  const int mode = env_int("GOCTR_DP_CAPTURE_COMM", 1);
  if (mode == 0 || !comm_capturable() || m->emb_lr > 0.f) return false;
  // (the self-test is a collective: it runs where every rank is known to be -- goctr_comm_init, or the start of a multi-device
  // call -- never lazily here, where a rank that happens to step eagerly would not take part)
  return mode == 2 || engine().capture_state == 1;
}

// kMulti[z] (even) consecutive steps starting at either parity as one graph each.  Without a communicator nothing splits the
// step; with one (dp_capture_ok) the all-reduce is a node of the graph: reduce | ncclAllReduce | Adam (+ the next step's
// attention when pipelined) -- a step inside a call costs no host-issued item at all instead of two
int build_multi_graphs(goctr_model* m, const RowSource& src, int B, const StepOpts& o) {
  Engine& e = engine();
  StepGraph& sg = m->graph;
  const int stp_now = m->stp;
  const bool dp = e.comm_active();
  const bool fuse = !dp;
  for (int z = 0; z < StepGraph::kNMulti; ++z)
    for (int par = 0; par < 2 && sg.kMulti[z] >= 2; ++par) {
      m->stp = par;
      const int rcg = capture_graph(e.stream, &sg.multi[z][par], [&] {
        int rc = 0;
        for (int k = 0; k < sg.kMulti[z] && !rc; ++k) {   // launch_backward flips m->stp: the captured steps alternate
          rc = launch_forward(m, src, B, o) || launch_backward(m, src, B, o, true, fuse);
          if (!rc && dp) rc = allreduce_grads(m) || launch_adam_step(m, src, B, o);
          else if (!rc && !fuse) rc = launch_adam(m, B, *o.tc);
        }
        return rc;
      }, [&] { m->stp = par; });
      m->stp = stp_now;
      if (rcg) return -1;
    }
  sg.multi_on = true;
  return 0;
}

int set_state(goctr_model* m, unsigned gstep, unsigned slot, long long batch_idx, long long n_batches) {
  StepState s{gstep, slot, batch_idx, n_batches};
  m->pend_retarget = false;
  if (m->ra_flag.p) GOCTR_HIP(hipMemsetAsync(m->ra_flag.p, 0, sizeof(unsigned int), engine().stream));   // (gstep may jump: no stale match)
  GOCTR_HIP(hipMemcpyAsync(m->st_cur(), &s, sizeof s, hipMemcpyHostToDevice, engine().stream));
  GOCTR_HIP(hipStreamSynchronize(engine().stream));
  return 0;
}

// point the running state at another batch of another dataset without a host round trip (gstep stays on the device), and
// give the state a call starts from its Adam bias corrections (ctr_kernels.h: StepState::corr1/2)
__global__ void step_state_prepare_kernel(StepState* st, double beta1, double beta2, int retarget, long long batch_idx, long long n_batches) {
  StepState s = *st;
  if (retarget) { s.slot = 0; s.batch_idx = batch_idx; s.n_batches = n_batches; }
  state_corrections(s, beta1, beta2);
  *st = s;
}
// (applied by run_steps' state-preparation launch: no kernel of its own)
int retarget_state(goctr_model* m, long long batch_idx, long long n_batches) {
  m->pend_retarget = true; m->pend_batch_idx = batch_idx; m->pend_n_batches = n_batches;
  return 0;
}

int get_state(goctr_model* m, StepState* s) {
  // (data parallel: the steps queued so far hold collectives -- wait for them under the communicator's watchdog, so that a peer
  // that failed makes this rank's call fail instead of blocking in the copy below)
  if (engine().comm_active() && comm_watch_stream()) return -1;
  GOCTR_HIP(hipMemcpyAsync(s, m->st_cur(), sizeof *s, hipMemcpyDeviceToHost, engine().stream));
  GOCTR_HIP(hipStreamSynchronize(engine().stream));
  return 0;
}

StepOpts opts_from(const goctr_train_cfg* tc) {
  StepOpts o;
  o.tc = tc; o.drop_mode = tc->dropout_mode; o.p0 = tc->p0; o.p1 = tc->p1; o.seed = tc->seed;
  return o;
}

int check_dataset(const goctr_model* m, const goctr_dataset* d, const goctr_emb* e) {
  const goctr_ctr_cfg& c = m->cfg;
  if (d->id_mode) {
    GOCTR_CHECK(e != nullptr, "id-mode dataset needs an embedding table");
    GOCTR_CHECK(e->D == c.D, "embedding dim %d != model D %d", e->D, c.D);
    GOCTR_CHECK(d->U == c.U && d->C == c.C && d->T == c.T, "dataset dims (U=%d,T=%d,C=%d) != model (U=%d,T=%d,C=%d)",
                d->U, d->T, d->C, c.U, c.T, c.C);
  } else {
    const int* r = d->ranges;
    GOCTR_CHECK(r[1] - r[0] == c.U && r[3] - r[2] == c.T * c.D && r[5] - r[4] == c.D && r[7] - r[6] == c.C,
                "SampleInfo ranges do not match the model dims");
    GOCTR_CHECK(r[7] <= d->xcols && r[0] >= 0, "SampleInfo ranges exceed xcols");
  }
  return 0;
}

// behind the last queued launch that writes the weights: what a serving slot's stream waits for (serve_wait_weights)
int mark_weights_written(goctr_model* m) {
  if (!m->ev_weights) GOCTR_HIP(hipEventCreateWithFlags(&m->ev_weights, hipEventDisableTiming));
  GOCTR_HIP(hipEventRecord(m->ev_weights, engine().stream));
  m->weights_pending.store(true, std::memory_order_release);
  return 0;
}

// behind the last queued launch that writes the table's rows (embedding training): serve_wait_rows
int emb_mark_written(goctr_emb* e) {
  if (!e->ev_rows) GOCTR_HIP(hipEventCreateWithFlags(&e->ev_rows, hipEventDisableTiming));
  GOCTR_HIP(hipEventRecord(e->ev_rows, engine().stream));
  e->rows_pending.store(true, std::memory_order_release);
  return 0;
}

// W0[U:U+2D,:]^T for the dpv GEMM of the plan path: built here (outside any capture), then maintained by the Adam kernels
int ensure_w0pv(goctr_model* m) {
  if (m->w0pv_live) return 0;
  const goctr_ctr_cfg& c = m->cfg;
  const int Np = round_up(2 * c.D, 16);
  hipLaunchKernelGGL(w0pv_transpose_kernel, dim3((unsigned)cdiv((long long)m->H1p * Np, 256)), dim3(256), 0, engine().stream, m->W.p, m->H1p,
                     c.U, 2 * c.D, Np, m->W0pvT.p);
  GOCTR_HIP(hipGetLastError());
  m->w0pv_live = true;
  m->graph.destroy();            // the captured Adam launches did not carry the pointer
  return 0;
}

// queue n_steps training steps (graph replay unless profiling / disabled)
int run_steps_impl(goctr_model* m, goctr_emb* emb, goctr_dataset* d, const goctr_train_cfg* tc, int n_steps);
int run_steps(goctr_model* m, goctr_emb* emb, goctr_dataset* d, const goctr_train_cfg* tc, int n_steps) {
  // A call that fails half way may already have queued launches that write the weights: the event is recorded on EVERY exit,
  // so a serving slot that takes the model's lock afterwards still waits for whatever was queued.
  const int rc = run_steps_impl(m, emb, d, tc, n_steps);
  if (n_steps > 0) {
    const std::string msg = rc ? goctr_last_error() : "";
    const int mrc = mark_weights_written(m);
    if (m->emb_lr > 0.f && emb) (void)emb_mark_written(emb);
    if (rc) { set_error("%s", msg.c_str()); return -1; }
    if (mrc) return -1;
  }
  return rc;
}
int run_steps_impl(goctr_model* m, goctr_emb* emb, goctr_dataset* d, const goctr_train_cfg* tc, int n_steps) {
  Engine& e = engine();
  const int B = tc->batch;
  if (ensure_workspace(m, B)) return -1;
  RowSource src = make_source(d, emb);
  StepOpts o = opts_from(tc);
  if (m->emb_lr > 0.f) {
    GOCTR_CHECK(src.id_mode, "embedding training needs an id-mode dataset (the dense TrainSample rows carry no ids)");
    if (ensure_emb_workspace(m, src.V, B)) return -1;
    if (emb_plan_ok(m, B) && emb_plan_fits(m, d, src.V, B)) { if (ensure_emb_plan(m, d, src, B) || ensure_w0pv(m)) return -1; }
    else { m->plan.valid = false; m->w0pv_live = false; }
  }
  const bool have_start = m->pend_retarget;                    // the host knows the batch the call starts at
  const long long start_batch = m->pend_batch_idx, start_nb = m->pend_n_batches;
  // (with a communicator and NO plan the sparse embedding exchange sizes its collectives from device counters read back by
  // the host: eager steps.  With the plan's fixed-size buckets the step is three captured graphs around the collectives.)
  const bool use_graph = !e.prof && env_int("GOCTR_NO_GRAPH", 0) == 0 && !(e.comm_active() && m->emb_lr > 0.f && !emb_split3(m));
  if (use_graph) o.pipelined = pipeline_ok(m, src);
  // The previous call ended exactly where this one starts and nothing happened in between (goctr_model::H0Carry): its last
  // launch computed this call's first h0 / gates, and its last loss block left the state this call starts from -- cursor,
  // Adam's bias corrections and all.
  const goctr_model::H0Carry& cy = m->carry;
  const bool retargeted = have_start;
  const bool carried = use_graph && o.pipelined && n_steps > 0 && cy.valid && retargeted && cy.gen + 1 == m->gen && cy.ds_uid == d->uid &&
                       emb && cy.emb_uid == emb->uid && cy.emb_version == emb->version && cy.B == B && cy.stp == m->stp &&
                       cy.batch == start_batch && cy.beta1 == (double)o.tc->beta1 && cy.beta2 == (double)o.tc->beta2 &&
                       cy.fac == gate_fac_mode(m, src, o, B) &&
                       env_int("GOCTR_H0_CARRY", 1) != 0;
  // The state-preparation launch: the cursor retarget of goctr_train_steps + the bias corrections of the state the call starts
  // from (ctr_kernels.h: StepState::corr1/2; later states get theirs from the loss block of the step before them).  A carried
  // start needs neither -- only the cost ring would not restart at slot 0, which matters to a caller that reads the costs.
  if (!(carried && m->pend_no_costs)) {
    hipLaunchKernelGGL(step_state_prepare_kernel, dim3(1), dim3(1), 0, e.stream, m->st_cur(), o.tc->beta1, o.tc->beta2,
                       m->pend_retarget ? 1 : 0, m->pend_batch_idx, m->pend_n_batches);
    GOCTR_HIP(hipGetLastError());
  }
  m->pend_retarget = false;
  if (use_graph) {
    if (!graph_matches(m->graph, d, emb, B, o, gate_fac_mode(m, src, o, B)) && build_graph(m, d, emb, src, B, o)) return -1;
    if (o.pipelined && n_steps > 0 && !carried) {
      // the first step's h0 (every later step gets it from its predecessor's last launch)
      const AttnArgs aa = make_attn_args(m, src, B, m->st_cur(), m->stp, gate_fac_mode(m, src, o, B));
      if (launch_attn_fwd(aa)) return -1;
    }
    m->carry.valid = false;
    if (m->emb_lr > 0.f && emb && n_steps > 0) ++emb->version;       // (rows are about to change: other models' carried h0 die)
    int s = 0;
    if ((!e.comm_active() || dp_capture_ok(m)) && env_int("GOCTR_GRAPH_STEPS", 1) != 0) {
      if (!m->graph.multi_on && build_multi_graphs(m, src, B, o)) return -1;
      // (long graphs first: a short one in front was measured slower at 20 steps per call, 66 vs 63.5 us per step)
      for (int z = 0; z < StepGraph::kNMulti; ++z) {   // even step counts: the parity is the same after each launch
        const int sz = m->graph.kMulti[z];
        for (; sz >= 2 && s + sz <= n_steps; s += sz) GOCTR_HIP(hipGraphLaunch(m->graph.multi[z][m->stp], e.stream));
      }
    }
    if (e.comm_active() && m->graph.ba[0] && m->graph.ba[1] && s < n_steps) {
      // dense data parallel: a(0) | all-reduce | [b(0) a(1)] | all-reduce | ... | [b(n-2) a(n-1)] | all-reduce | b(n-1)
      int par = m->stp;
      GOCTR_HIP(hipGraphLaunch(m->graph.a[par], e.stream));
      for (; s < n_steps; ++s) {
        m->stp ^= 1;
        if (allreduce_grads(m)) return -1;
        if (s + 1 < n_steps) { GOCTR_HIP(hipGraphLaunch(m->graph.ba[par], e.stream)); par ^= 1; }
        else GOCTR_HIP(hipGraphLaunch(m->graph.b[par], e.stream));
      }
    }
    for (; s < n_steps; ++s) {
      const int par = m->stp;
      GOCTR_HIP(hipGraphLaunch(m->graph.a[par], e.stream));
      if (m->graph.mid[par]) {          // data parallel + trainable embeddings: all-to-all, owner side + slab reduce, all-gather
        if (emb_exchange_a2a(m)) return -1;
        GOCTR_HIP(hipGraphLaunch(m->graph.mid[par], e.stream));
        if (emb_exchange_gather(m)) return -1;
      }
      m->stp ^= 1;
      if (e.comm_active()) {
        if (allreduce_grads(m)) return -1;
        GOCTR_HIP(hipGraphLaunch(m->graph.b[par], e.stream));
      }
    }
    if (o.pipelined && n_steps > 0 && retargeted && emb && start_nb > 0) {
      m->carry = goctr_model::H0Carry{true, m->gen, d->uid, emb->uid, emb->version, B, m->stp, (start_batch + n_steps) % start_nb,
                                      (double)o.tc->beta1, (double)o.tc->beta2, gate_fac_mode(m, src, o, B)};
    }
  } else {
    if (m->emb_lr > 0.f && emb && n_steps > 0) ++emb->version;
    m->carry.valid = false;
    // GOCTR_EAGER_PIPELINE=1 (profiling: the rocprofv3 counter passes want ONE dispatch record per launch AND the kernels of
    // the replayed step): the pipelined launch sequence -- chain, weight gradients, reduce_attn with the next step's attention --
    // issued eagerly, launch by launch, instead of as a captured graph
    if (!e.prof && !e.comm_active() && n_steps > 0 && env_int("GOCTR_EAGER_PIPELINE", 0) != 0 && pipeline_ok(m, src)) {
      o.pipelined = true;
      const AttnArgs aa = make_attn_args(m, src, B, m->st_cur(), m->stp, gate_fac_mode(m, src, o, B));
      if (launch_attn_fwd(aa)) return -1;
    }
    for (int s = 0; s < n_steps; ++s)
      if (train_step_eager(m, src, B, o)) return -1;
  }
  return 0;
}

int upload_padded_weights(goctr_model* m, int tensor_id, const float* host, size_t n) {
  const goctr_ctr_cfg& c = m->cfg;
  std::vector<float> buf;
  Engine& e = engine();
  auto up = [&](float* dst, const std::vector<float>& v) -> int {
    GOCTR_HIP(hipMemcpyAsync(dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice, e.stream));
    GOCTR_HIP(hipStreamSynchronize(e.stream));
    return 0;
  };
  switch (tensor_id) {
    case GOCTR_W0: {
      GOCTR_CHECK(n == (size_t)m->I * c.H1, "W0 expects %d floats, got %zu", m->I * c.H1, n);
      buf.assign((size_t)m->Ip * m->H1p, 0.f);
      for (int r = 0; r < m->I; ++r) for (int k = 0; k < c.H1; ++k) buf[(size_t)r * m->H1p + k] = host[(size_t)r * c.H1 + k];
      if (up(m->W.p, buf)) return -1;
      std::vector<float> t((size_t)m->H1p * m->Dp, 0.f), ti((size_t)m->H1p * m->Dp, 0.f), wi((size_t)m->Ip * m->H1p, 0.f);
      for (int d = 0; d < c.D; ++d) for (int k = 0; k < c.H1; ++k) {
        t[(size_t)k * m->Dp + d] = host[(size_t)(c.U + d) * c.H1 + k];
        ti[img_index(k, d, m->Dp)] = host[(size_t)(c.U + d) * c.H1 + k];
      }
      for (int r = 0; r < m->I; ++r) for (int k = 0; k < c.H1; ++k) wi[img_index(r, k, m->H1p)] = host[(size_t)r * c.H1 + k];
      if (up(m->img(0), wi) || up(m->img(3), ti)) return -1;
      return up(m->W0sT.p, t);
    }
    case GOCTR_W1: {
      GOCTR_CHECK(n == (size_t)c.H1 * c.H2, "W1 expects %d floats, got %zu", c.H1 * c.H2, n);
      buf.assign((size_t)m->H1p * m->H2p, 0.f);
      std::vector<float> t((size_t)m->H2p * m->H1p, 0.f);
      for (int r = 0; r < c.H1; ++r) for (int k = 0; k < c.H2; ++k) {
        buf[(size_t)r * m->H2p + k] = host[(size_t)r * c.H2 + k];
        t[(size_t)k * m->H1p + r] = host[(size_t)r * c.H2 + k];
      }
      std::vector<float> wi((size_t)m->H1p * m->H2p, 0.f), ti((size_t)m->H2p * m->H1p, 0.f);
      for (int r = 0; r < c.H1; ++r) for (int k = 0; k < c.H2; ++k) {
        wi[img_index(r, k, m->H2p)] = host[(size_t)r * c.H2 + k];
        ti[img_index(k, r, m->H1p)] = host[(size_t)r * c.H2 + k];
      }
      if (up(m->W.p + m->off1, buf) || up(m->img(1), wi) || up(m->img(2), ti)) return -1;
      return up(m->W1T.p, t);
    }
    case GOCTR_W2: {
      GOCTR_CHECK(n == (size_t)c.H2, "W2 expects %d floats, got %zu", c.H2, n);
      buf.assign((size_t)m->H2p * 16, 0.f);
      std::vector<float> t((size_t)16 * m->H2p, 0.f);
      for (int r = 0; r < c.H2; ++r) { buf[(size_t)r * 16] = host[r]; t[r] = host[r]; }
      if (up(m->W.p + m->off2, buf)) return -1;
      return up(m->W2T.p, t);
    }
    case GOCTR_ATT0: {
      GOCTR_CHECK(n == (size_t)c.T, "att0 expects %d floats, got %zu", c.T, n);
      buf.assign((size_t)m->Tp, 0.f);
      for (int t = 0; t < c.T; ++t) buf[t] = host[t];
      return up(m->W.p + m->offa, buf);
    }
  }
  set_error("unknown tensor id %d", tensor_id);
  return -1;
}

int download_padded(goctr_model* m, const float* flat_dev, int tensor_id, float* host, size_t n) {
  const goctr_ctr_cfg& c = m->cfg;
  Engine& e = engine();
  std::vector<float> buf;
  auto down = [&](const float* src, size_t cnt) -> int {
    buf.resize(cnt);
    GOCTR_HIP(hipMemcpyAsync(buf.data(), src, cnt * sizeof(float), hipMemcpyDeviceToHost, e.stream));
    GOCTR_HIP(hipStreamSynchronize(e.stream));
    return 0;
  };
  switch (tensor_id) {
    case GOCTR_W0:
      GOCTR_CHECK(n == (size_t)m->I * c.H1, "W0 expects %d floats, got %zu", m->I * c.H1, n);
      if (down(flat_dev, (size_t)m->Ip * m->H1p)) return -1;
      for (int r = 0; r < m->I; ++r) for (int k = 0; k < c.H1; ++k) host[(size_t)r * c.H1 + k] = buf[(size_t)r * m->H1p + k];
      return 0;
    case GOCTR_W1:
      GOCTR_CHECK(n == (size_t)c.H1 * c.H2, "W1 expects %d floats, got %zu", c.H1 * c.H2, n);
      if (down(flat_dev + m->off1, (size_t)m->H1p * m->H2p)) return -1;
      for (int r = 0; r < c.H1; ++r) for (int k = 0; k < c.H2; ++k) host[(size_t)r * c.H2 + k] = buf[(size_t)r * m->H2p + k];
      return 0;
    case GOCTR_W2:
      GOCTR_CHECK(n == (size_t)c.H2, "W2 expects %d floats, got %zu", c.H2, n);
      if (down(flat_dev + m->off2, (size_t)m->H2p * 16)) return -1;
      for (int r = 0; r < c.H2; ++r) host[r] = buf[(size_t)r * 16];
      return 0;
    case GOCTR_ATT0:
      GOCTR_CHECK(n == (size_t)c.T, "att0 expects %d floats, got %zu", c.T, n);
      if (down(flat_dev + m->offa, (size_t)m->Tp)) return -1;
      for (int t = 0; t < c.T; ++t) host[t] = buf[t];
      return 0;
  }
  set_error("unknown tensor id %d", tensor_id);
  return -1;
}

}  // namespace

extern "C" {

void goctr_train_cfg_default(goctr_train_cfg* c) {
  memset(c, 0, sizeof *c);
  c->batch = 200; c->epochs = 200; c->early_stop = 20;  // dinimpl_test.go:36-43
  c->lr = 0.01; c->l2 = 0.0001;                           // model.go:88
  c->beta1 = 0.9; c->beta2 = 0.999; c->eps = 1e-8;
  c->adam_div_by_batch = 1; c->adam_l2_before_batch_div = 1;
  c->dropout_mode = 2; c->p0 = 0.005f; c->p1 = 0.005f; c->seed = 42;   // din.go:204-205,307-312: Dropout is always on
}

int goctr_model_create(const goctr_ctr_cfg* cfg, goctr_model** out) {
  GOCTR_ENTER();
  GOCTR_CHECK(cfg && out, "goctr_model_create: null argument");
  GOCTR_CHECK(cfg->kind == GOCTR_DIN || cfg->kind == GOCTR_YOUTUBE, "unknown model kind %d", cfg->kind);
  GOCTR_CHECK(cfg->U >= 0 && cfg->T > 0 && cfg->D > 0 && cfg->C >= 0 && cfg->H1 > 0 && cfg->H2 > 0, "bad model dims");
  GOCTR_CHECK(cfg->D <= 256, "embedding dim %d > 256 not supported", cfg->D);
  if (init_kernel_attrs()) return -1;
  std::unique_ptr<goctr_model> m(new goctr_model);
  m->cfg = *cfg;
  m->I = cfg->U + 2 * cfg->D + cfg->C;
  m->Ip = round_up(m->I, 16); m->H1p = round_up(cfg->H1, 16); m->H2p = round_up(cfg->H2, 16);
  m->Dp = round_up(cfg->D, 16); m->Tp = round_up(cfg->T, 16);
  m->off1 = m->Ip * m->H1p; m->off2 = m->off1 + m->H1p * m->H2p; m->offa = m->off2 + m->H2p * 16;
  m->nflat = m->offa + m->Tp;
  if (m->W.alloc(m->nflat) || m->G.alloc((size_t)m->nflat + 1) || m->Mo.alloc(m->nflat) || m->Vo.alloc(m->nflat)) return -1;
  if (m->W1T.alloc((size_t)m->H2p * m->H1p) || m->W2T.alloc((size_t)16 * m->H2p) || m->W0sT.alloc((size_t)m->H1p * m->Dp)) return -1;
  if (m->Wimg.alloc((size_t)m->off1 + 2 * (size_t)m->H1p * m->H2p + (size_t)m->H1p * m->Dp)) return -1;
  if (chain_x3_shape_ok(m.get())) {
    m->x3_nch0 = m->Ip / 16;
    if (m->Wx3.alloc(cx_images_elems(m->x3_nch0))) return -1;      // zero = the images of all-zero weights
  }
  if (m->st.alloc(2) || m->costs.alloc(COST_RING)) return -1;
  std::vector<float> ones(cfg->T, 1.0f);  // din.go:181 att0 = 1
  if (upload_padded_weights(m.get(), GOCTR_ATT0, ones.data(), ones.size())) return -1;
  if (set_state(m.get(), 0, 0, 0, 1)) return -1;
  *out = m.release();
  return 0;
}

void goctr_model_destroy(goctr_model* m) {
  if (!m) return;
  for (goctr_model* r : m->reps) goctr_model_destroy(r);        // (replicas of the multi-device entry, on their own engines)
  m->reps.clear();
  EngineScope on(m->eng);
  std::lock_guard<std::recursive_mutex> lk(m->eng->mu);
  // (the engine's own streams, not hipDeviceSynchronize: a device-wide wait invalidates the stream capture of any OTHER thread
  // that is building its step graphs on this device -- a second logical rank, or a training goroutine beside a serving one;
  // serving passes are synchronous, none of this model's is in flight once its caller returned)
  if (engine().inited) { (void)hipStreamSynchronize(engine().stream); (void)hipStreamSynchronize(engine().side); }
  m->graph.destroy();
  if (m->ev_weights) (void)hipEventDestroy(m->ev_weights);
  delete m;
}

int goctr_model_set_weights(goctr_model* m, int tensor_id, const float* host, size_t n) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m && host, "goctr_model_set_weights: null argument");
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  if (upload_padded_weights(m, tensor_id, host, n)) return -1;
  if (tensor_id == GOCTR_W0) m->w0pv_live = false;
  if ((tensor_id == GOCTR_W0 || tensor_id == GOCTR_W1) && rebuild_x3_images(m)) return -1;
  return mark_weights_written(m);
}

int goctr_model_get_weights(goctr_model* m, int tensor_id, float* host, size_t n) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m && host, "goctr_model_get_weights: null argument");
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  return download_padded(m, m->W.p, tensor_id, host, n);
}

// Adam moments of one tensor in the tensor's own (unpadded, row-major) shape: what a checkpoint needs next to the
// weights to resume `model.Train` where it stopped.  which = 0: first moment, 1: second moment.
int upload_padded_flat(goctr_model* m, float* flat_dev, int tensor_id, const float* host, size_t n) {
  const goctr_ctr_cfg& c = m->cfg;
  std::vector<float> buf;
  size_t off = 0;
  switch (tensor_id) {
    case GOCTR_W0:
      GOCTR_CHECK(n == (size_t)m->I * c.H1, "W0 expects %d floats, got %zu", m->I * c.H1, n);
      buf.assign((size_t)m->Ip * m->H1p, 0.f);
      for (int r = 0; r < m->I; ++r) for (int k = 0; k < c.H1; ++k) buf[(size_t)r * m->H1p + k] = host[(size_t)r * c.H1 + k];
      break;
    case GOCTR_W1:
      GOCTR_CHECK(n == (size_t)c.H1 * c.H2, "W1 expects %d floats, got %zu", c.H1 * c.H2, n);
      buf.assign((size_t)m->H1p * m->H2p, 0.f); off = m->off1;
      for (int r = 0; r < c.H1; ++r) for (int k = 0; k < c.H2; ++k) buf[(size_t)r * m->H2p + k] = host[(size_t)r * c.H2 + k];
      break;
    case GOCTR_W2:
      GOCTR_CHECK(n == (size_t)c.H2, "W2 expects %d floats, got %zu", c.H2, n);
      buf.assign((size_t)m->H2p * 16, 0.f); off = m->off2;
      for (int r = 0; r < c.H2; ++r) buf[(size_t)r * 16] = host[r];
      break;
    case GOCTR_ATT0:
      GOCTR_CHECK(n == (size_t)c.T, "att0 expects %d floats, got %zu", c.T, n);
      buf.assign((size_t)m->Tp, 0.f); off = m->offa;
      for (int t = 0; t < c.T; ++t) buf[t] = host[t];
      break;
    default:
      GOCTR_CHECK(false, "unknown tensor id %d", tensor_id);
  }
  GOCTR_HIP(hipMemcpyAsync(flat_dev + off, buf.data(), buf.size() * sizeof(float), hipMemcpyHostToDevice, engine().stream));
  GOCTR_HIP(hipStreamSynchronize(engine().stream));
  return 0;
}

int goctr_model_get_moments(goctr_model* m, int tensor_id, int which, float* host, size_t n) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m && host && (which == 0 || which == 1), "goctr_model_get_moments: bad argument");
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  return download_padded(m, which ? m->Vo.p : m->Mo.p, tensor_id, host, n);
}

int goctr_model_set_moments(goctr_model* m, int tensor_id, int which, const float* host, size_t n) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m && host && (which == 0 || which == 1), "goctr_model_set_moments: bad argument");
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  return upload_padded_flat(m, which ? m->Vo.p : m->Mo.p, tensor_id, host, n);
}

// Global step counter: Adam's iteration number and the dropout stream position.
int goctr_model_get_step(goctr_model* m, uint32_t* step) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m && step, "goctr_model_get_step: null argument");
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  StepState s;
  if (get_state(m, &s)) return -1;
  *step = s.gstep;
  return 0;
}

int goctr_model_set_step(goctr_model* m, uint32_t step) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m, "goctr_model_set_step: null argument");
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  return set_state(m, step, 0, 0, 1);
}

int goctr_model_set_embedding_training(goctr_model* m, double lr) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m && lr >= 0 && lr == lr, "goctr_model_set_embedding_training: bad arguments");
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  GOCTR_CHECK(lr == 0 || m->cfg.D <= 64, "embedding training supports D <= 64 (got %d)", m->cfg.D);
  m->emb_lr = (float)lr;
  m->w0pv_live = false;            // (the Adam kernels only keep W0pvT current while embedding training is on)
  m->graph.destroy();
  return 0;
}

int goctr_model_get_emb_plan(goctr_model* m, int64_t* n_batches, int64_t* n_pairs, int64_t* n_slots, int32_t* pair, int32_t* pslot,
                             int32_t* pid, int32_t* slot_id, uint32_t* slot_off, int64_t* pair_off, int64_t* slot_base) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m, "goctr_model_get_emb_plan: null argument");
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  const auto& P = m->plan;
  GOCTR_CHECK(P.valid, "goctr_model_get_emb_plan: no plan resident (run an embedding-training step first)");
  if (n_batches) *n_batches = P.nb;
  if (n_pairs) *n_pairs = P.total_pairs;
  if (n_slots) *n_slots = P.total_slots;
  const size_t np = (size_t)P.total_pairs, ns = (size_t)P.total_slots, nb = (size_t)P.nb;
  if (pair && np && P.pair.download(pair, np)) return -1;
  if (pslot && np && P.pslot.download(pslot, np)) return -1;
  if (pid && np && P.pid.download(pid, np)) return -1;
  if (slot_id && ns && P.slot_id.download(slot_id, ns)) return -1;
  if (slot_off && P.slot_off.download(slot_off, ns + nb)) return -1;
  if (pair_off && P.pair_off.download(reinterpret_cast<long long*>(pair_off), nb + 1)) return -1;
  if (slot_base && P.slot_base.download(reinterpret_cast<long long*>(slot_base), nb + 1)) return -1;
  return 0;
}

int goctr_model_emb_plan_build_ms(goctr_model* m, double* ms, int64_t* n_batches) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m, "goctr_model_emb_plan_build_ms: null argument");
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  GOCTR_CHECK(m->plan.valid, "goctr_model_emb_plan_build_ms: no plan resident (run an embedding-training step first)");
  if (ms) *ms = m->plan.build_ms;
  if (n_batches) *n_batches = m->plan.nb;
  return 0;
}

int goctr_model_sparse_exchange_bytes(goctr_model* m, double* bytes) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m && bytes, "goctr_model_sparse_exchange_bytes: null argument");
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  *bytes = m->emb_comm ? m->ex_bytes_last : 0.0;      // (emb_comm: the last step's sparse update ran with a communicator)
  return 0;
}

int goctr_emb_get_rows(goctr_emb* e, int64_t first, int64_t n, float* host_rows) {
  GOCTR_ENTER_H(e);
  GOCTR_CHECK(e && host_rows && first >= 0 && n >= 0 && first + n <= e->V, "goctr_emb_get_rows: range out of bounds");
  return n ? e->rows.download(host_rows, (size_t)n * e->D, (size_t)first * e->D) : 0;
}

int goctr_model_reset_optimizer(goctr_model* m) {
  GOCTR_ENTER_H(m);
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  GOCTR_HIP(hipMemsetAsync(m->Mo.p, 0, sizeof(float) * m->nflat, engine().stream));
  GOCTR_HIP(hipMemsetAsync(m->Vo.p, 0, sizeof(float) * m->nflat, engine().stream));
  return set_state(m, 0, 0, 0, 1);
}

// ------------------------------------------------------------------ embedding table / gather
int goctr_emb_create(int64_t V, int D, const float* host_rows, goctr_emb** out) {
  GOCTR_ENTER();
  GOCTR_CHECK(V > 0 && D > 0 && out, "goctr_emb_create: bad arguments");
  std::unique_ptr<goctr_emb> e(new goctr_emb);
  e->V = V; e->D = D;
  if (e->rows.alloc((size_t)(V + 1) * D)) return -1;   // row V stays all-zero: where missing ids point (attention kernels)
  if (host_rows && e->rows.upload(host_rows, (size_t)V * D)) return -1;
  *out = e.release();
  return 0;
}
int goctr_emb_set_rows(goctr_emb* e, int64_t first, int64_t n, const float* host_rows) {
  GOCTR_ENTER_H(e);
  GOCTR_CHECK(e && host_rows && first >= 0 && n >= 0 && first + n <= e->V, "goctr_emb_set_rows: range out of bounds");
  std::unique_lock<std::shared_mutex> lk(e->mu);       // (serving passes read the rows under the shared lock; the upload below is synchronous)
  ++e->version;
  return n ? e->rows.upload(host_rows, (size_t)n * e->D, (size_t)first * e->D) : 0;
}
void goctr_emb_destroy(goctr_emb* e) {
  if (!e) return;
  for (goctr_emb* r : e->reps) goctr_emb_destroy(r);
  e->reps.clear();
  EngineScope on(e->eng);
  std::lock_guard<std::recursive_mutex> lk(e->eng->mu);
  if (engine().inited) (void)hipStreamSynchronize(engine().stream);
  if (e->ev_rows) (void)hipEventDestroy(e->ev_rows);
  delete e;
}

int goctr_gather_rows(goctr_emb* e, const int32_t* ub_ids, const int32_t* item_ids, const float* user_feat, int U,
                      const float* ctx_feat, int C, int T, int64_t rows, float* X_out) {
  GOCTR_ENTER_H(e);
  GOCTR_CHECK(e && X_out && rows >= 0, "goctr_gather_rows: bad arguments");
  if (rows == 0) return 0;
  const int xcols = U + T * e->D + e->D + C;
  DevBuf<int32_t> dub, dit; DevBuf<float> duf, dcf, dX;
  if (dub.alloc((size_t)rows * T, false) || dit.alloc(rows, false) || duf.alloc((size_t)rows * U, false) ||
      dcf.alloc((size_t)rows * C, false) || dX.alloc((size_t)rows * xcols, false)) return -1;
  if (dub.upload(ub_ids, (size_t)rows * T) || dit.upload(item_ids, rows)) return -1;
  if (U && duf.upload(user_feat, (size_t)rows * U)) return -1;
  if (C && dcf.upload(ctx_feat, (size_t)rows * C)) return -1;
  GatherArgs a{e->rows.p, e->V, e->D, T, U, C, dub.p, dit.p, duf.p, dcf.p, rows, dX.p, xcols};
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, engine().stream, a);
  GOCTR_HIP(hipGetLastError());
  return dX.download(X_out, (size_t)rows * xcols);
}

// ------------------------------------------------------------------ datasets
int goctr_dataset_create_dense(const float* X, const float* Y, int64_t rows, int xcols, const int ranges[8],
                               goctr_dataset** out) {
  GOCTR_ENTER();
  GOCTR_CHECK(X && rows > 0 && xcols > 0 && ranges && out, "goctr_dataset_create_dense: bad arguments");
  std::unique_ptr<goctr_dataset> d(new goctr_dataset);
  d->id_mode = false; d->rows = rows; d->xcols = xcols;
  memcpy(d->ranges, ranges, sizeof d->ranges);
  if (d->X.alloc((size_t)rows * xcols, false) || d->X.upload(X, (size_t)rows * xcols)) return -1;
  if (Y) { if (d->Y.alloc(rows, false) || d->Y.upload(Y, rows)) return -1; d->has_y = true; }
  *out = d.release();
  return 0;
}

int goctr_dataset_create_ids(const int32_t* ub_ids, const int32_t* item_ids, const float* user_feat, int U,
                             const float* ctx_feat, int C, int T, const float* Y, int64_t rows, goctr_dataset** out) {
  GOCTR_ENTER();
  GOCTR_CHECK(ub_ids && item_ids && rows > 0 && T > 0 && out, "goctr_dataset_create_ids: bad arguments");
  std::unique_ptr<goctr_dataset> d(new goctr_dataset);
  d->id_mode = true; d->rows = rows; d->U = U; d->C = C; d->T = T;
  if (d->ub_ids.alloc((size_t)rows * T, false) || d->ub_ids.upload(ub_ids, (size_t)rows * T)) return -1;
  if (d->item_ids.alloc(rows, false) || d->item_ids.upload(item_ids, rows)) return -1;
  if (d->ufeat.alloc((size_t)rows * U, false) || (U && d->ufeat.upload(user_feat, (size_t)rows * U))) return -1;
  if (d->cfeat.alloc((size_t)rows * C, false) || (C && d->cfeat.upload(ctx_feat, (size_t)rows * C))) return -1;
  if (Y) { if (d->Y.alloc(rows, false) || d->Y.upload(Y, rows)) return -1; d->has_y = true; }
  *out = d.release();
  return 0;
}
void goctr_dataset_destroy(goctr_dataset* d) {
  if (!d) return;
  for (goctr_dataset* s : d->shards) goctr_dataset_destroy(s);
  d->shards.clear();
  EngineScope on(d->eng);
  std::lock_guard<std::recursive_mutex> lk(d->eng->mu);
  if (engine().inited) (void)hipStreamSynchronize(engine().stream);   // queued (asynchronous) steps may still read the rows
  delete d;
}

}  // extern "C"

// ------------------------------------------------------------------ device-side sample assembly
// ubcache.UserBehaviorCache (feature/ubcache/cache.go) as a CSR in HBM + the per-sample gather of GetSampleVector
// (recommend/rcmd.go:460-536) as one kernel: keys (user, item, timestamp) -> behaviour ids, user / item feature rows.
struct goctr_ubcache {
  goctr::Engine* const eng = &goctr::engine();   // the engine (device, streams, arena) the handle was created on
  int64_t n_users = 0, nnz = 0;
  DevBuf<long long> off, ts;
  DevBuf<int32_t> items;
};

namespace {
// TimeSeq.Filter (cache.go:71-94) for one key: the sequence is newest-first, so "the first i with Ts[i] <= maxTs"
// is a lower bound found by bisection; then up to T items from there.
__global__ __launch_bounds__(256) void assemble_keys_kernel(const long long* __restrict__ off, const int32_t* __restrict__ seq_items,
                                                            const long long* __restrict__ seq_ts, long long n_users,
                                                            const float* __restrict__ user_table, int U,
                                                            const float* __restrict__ item_table, long long n_items, int C,
                                                            const int32_t* __restrict__ users, const int32_t* __restrict__ items,
                                                            const long long* __restrict__ ts, long long rows, int T,
                                                            int32_t* __restrict__ ub_ids, float* __restrict__ ufeat,
                                                            float* __restrict__ cfeat, int32_t* __restrict__ item_out,
                                                            unsigned char* __restrict__ failed) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wavefront per sample
  if (r >= rows) return;
  // the key's three fields first, back to back: in a small serving pass they sit in pinned HOST memory (zero-copy), and
  // fetched one by one where they are used they were three PCIe round trips in a row (7.8 us for a 256-key pass)
  const int u = users[r];
  const int it_key = items ? items[r] : -1;
  const long long ts_key = ts ? ts[r] : 0;
  bool uok = u >= 0 && u < n_users;
  if (failed) {
    // BatchPredict (rcmd.go:291-307): a key whose GetUserFeature / GetItemFeature fails is scored as the ALL-zero row
    // (user features, behaviours, item embedding and item features alike)
    const int it = it_key;
    const bool ok = uok && it >= 0 && it < n_items;
    if (lane == 0) { failed[r] = ok ? 0 : 1; item_out[r] = ok ? it : -1; }
    if (!ok) {
      for (int j = lane; j < T; j += 64) ub_ids[r * T + j] = -1;
      for (int j = lane; j < U; j += 64) ufeat[r * U + j] = 0.f;
      for (int j = lane; j < C; j += 64) cfeat[r * C + j] = 0.f;
      return;
    }
  }
  long long first = 0, cnt = 0;
  const long long b = (uok && off) ? off[u] : 0, len = (uok && off) ? off[u + 1] - b : 0;   // off == NULL: no behaviour cache
  if (len > 0) {
    const long long mts = ts_key;
    // first i with seq_ts[b + i] <= mts (descending order); mts == 0 means "from the newest" (cache.go:72-74: maxTs = Ts[0])
    long long lo = 0;
    if (mts != 0) {
      if (len <= 256) {
        // short histories (the common case): 64 entries per coalesced load and one ballot instead of a chain of ~7 dependent
        // loads -- the serving pass of a Rank call is latency, not work
        lo = len;
        for (long long base = 0; base < len; base += 64) {
          const long long i = base + lane;
          const unsigned long long le = __ballot(i < len && seq_ts[b + i] <= mts);
          if (le) { lo = base + (long long)__builtin_ctzll(le); break; }
        }
      } else {
        long long hi = len;
        while (lo < hi) {
          const long long mid = (lo + hi) >> 1;
          if (seq_ts[b + mid] <= mts) hi = mid; else lo = mid + 1;
        }
      }
    }
    first = lo;
    cnt = len - first < T ? len - first : T;
  }
  if (ub_ids)
    for (int j = lane; j < T; j += 64) ub_ids[r * T + j] = j < cnt ? seq_items[b + first + j] : -1;
  if (ufeat)
    for (int j = lane; j < U; j += 64) ufeat[r * U + j] = uok ? user_table[(long long)u * U + j] : 0.f;
  if (cfeat) {
    const int it = it_key;
    const bool iok = it >= 0 && it < n_items;
    for (int j = lane; j < C; j += 64) cfeat[r * C + j] = iok ? item_table[(long long)it * C + j] : 0.f;
  }
}
}  // namespace

extern "C" {

int goctr_ubcache_create(int64_t n_users, const int64_t* off, const int32_t* items, const int64_t* ts, goctr_ubcache** out) {
  GOCTR_ENTER();
  GOCTR_CHECK(n_users > 0 && off && out && off[0] == 0, "goctr_ubcache_create: bad arguments");
  const int64_t nnz = off[n_users];
  GOCTR_CHECK(nnz >= 0 && (nnz == 0 || (items && ts)), "goctr_ubcache_create: sequences missing");
  for (int64_t u = 0; u < n_users; ++u) {
    GOCTR_CHECK(off[u + 1] >= off[u], "goctr_ubcache_create: offsets must be non-decreasing");
    for (int64_t k = off[u] + 1; k < off[u + 1]; ++k)
      GOCTR_CHECK(ts[k] <= ts[k - 1], "goctr_ubcache_create: user %lld's sequence is not in timestamp-descending order "
                  "(cache.go:8 TimeSeq)", (long long)u);
  }
  std::unique_ptr<goctr_ubcache> c(new goctr_ubcache);
  c->n_users = n_users; c->nnz = nnz;
  std::vector<long long> o(off, off + n_users + 1), t(ts, ts + nnz);
  if (c->off.alloc(o.size(), false) || c->off.upload(o.data(), o.size())) return -1;
  if (c->items.alloc((size_t)nnz, false) || (nnz && c->items.upload(items, (size_t)nnz))) return -1;
  if (c->ts.alloc((size_t)nnz, false) || (nnz && c->ts.upload(t.data(), (size_t)nnz))) return -1;
  *out = c.release();
  return 0;
}
void goctr_ubcache_destroy(goctr_ubcache* c) { delete c; }

int goctr_ubcache_get(goctr_ubcache* c, const int32_t* users, const int64_t* max_ts, int64_t rows, int T, int32_t* out_ids) {
  GOCTR_ENTER_H(c);
  GOCTR_CHECK(c && users && out_ids && rows > 0 && T > 0, "goctr_ubcache_get: bad arguments");
  DevBuf<int32_t> du, dout; DevBuf<long long> dts;
  std::vector<long long> t(rows, 0);
  if (max_ts) for (int64_t i = 0; i < rows; ++i) t[i] = max_ts[i];
  if (du.alloc(rows, false) || du.upload(users, rows) || dts.alloc(rows, false) || dts.upload(t.data(), rows) ||
      dout.alloc((size_t)rows * T, false)) return -1;
  hipLaunchKernelGGL(assemble_keys_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, engine().stream, c->off.p, c->items.p,
                     c->ts.p, (long long)c->n_users, (const float*)nullptr, 0, (const float*)nullptr, 0LL, 0, du.p,
                     (const int32_t*)nullptr, dts.p, (long long)rows, T, dout.p, (float*)nullptr, (float*)nullptr,
                     (int32_t*)nullptr, (unsigned char*)nullptr);
  GOCTR_HIP(hipGetLastError());
  return dout.download(out_ids, (size_t)rows * T);
}

int goctr_dataset_create_keys(goctr_ubcache* c, const float* user_table, int64_t n_users, int U, const float* item_table,
                              int64_t n_items, int C, const int32_t* users, const int32_t* items, const int64_t* ts,
                              const float* Y, int64_t rows, int T, goctr_dataset** out) {
  GOCTR_ENTER_H(c);
  GOCTR_CHECK(c && users && items && rows > 0 && T > 0 && out && n_items >= 0 && U >= 0 && C >= 0,
              "goctr_dataset_create_keys: bad arguments");
  GOCTR_CHECK(n_users == c->n_users, "goctr_dataset_create_keys: user table has %lld rows, the behaviour cache %lld users",
              (long long)n_users, (long long)c->n_users);
  GOCTR_CHECK((U == 0 || user_table) && (C == 0 || item_table), "goctr_dataset_create_keys: feature table missing");
  std::unique_ptr<goctr_dataset> d(new goctr_dataset);
  d->id_mode = true; d->rows = rows; d->U = U; d->C = C; d->T = T;
  DevBuf<float> dut, dit; DevBuf<int32_t> du; DevBuf<long long> dts;
  std::vector<long long> t(rows, 0);
  if (ts) for (int64_t i = 0; i < rows; ++i) t[i] = ts[i];
  if (dut.alloc((size_t)n_users * U, false) || (U && dut.upload(user_table, (size_t)n_users * U))) return -1;
  if (dit.alloc((size_t)n_items * C, false) || (C && dit.upload(item_table, (size_t)n_items * C))) return -1;
  if (du.alloc(rows, false) || du.upload(users, rows) || dts.alloc(rows, false) || dts.upload(t.data(), rows)) return -1;
  if (d->ub_ids.alloc((size_t)rows * T, false) || d->item_ids.alloc(rows, false) || d->item_ids.upload(items, rows)) return -1;
  if (d->ufeat.alloc((size_t)rows * U, false) || d->cfeat.alloc((size_t)rows * C, false)) return -1;
  hipLaunchKernelGGL(assemble_keys_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, engine().stream, c->off.p, c->items.p,
                     c->ts.p, (long long)c->n_users, dut.p, U, dit.p, (long long)n_items, C, du.p, d->item_ids.p, dts.p,
                     (long long)rows, T, d->ub_ids.p, d->ufeat.p, d->cfeat.p, (int32_t*)nullptr, (unsigned char*)nullptr);
  GOCTR_HIP(hipGetLastError());
  GOCTR_HIP(hipStreamSynchronize(engine().stream));   // the temporaries above are released on return
  if (Y) { if (d->Y.alloc(rows, false) || d->Y.upload(Y, rows)) return -1; d->has_y = true; }
  *out = d.release();
  return 0;
}

// read back the assembled keys of an id-mode dataset (tests, debugging)
int goctr_dataset_get_ids(goctr_dataset* d, int32_t* ub_ids, float* user_feat, float* ctx_feat) {
  GOCTR_ENTER_H(d);
  GOCTR_CHECK(d && d->id_mode, "goctr_dataset_get_ids: not an id-mode dataset");
  if (ub_ids && d->ub_ids.download(ub_ids, (size_t)d->rows * d->T)) return -1;
  if (user_feat && d->U && d->ufeat.download(user_feat, (size_t)d->rows * d->U)) return -1;
  if (ctx_feat && d->C && d->cfeat.download(ctx_feat, (size_t)d->rows * d->C)) return -1;
  return 0;
}

}  // extern "C"


// ------------------------------------------------------------------ single-call multi-device training (goctr_train_cfg::devices)
// recommend.Train -> Fitter.Fit -> model.Train is ONE call from ONE Go process (recommend/rcmd.go:196-246, model/model.go:27-213).
// After goctr_init_devices(n, ids) a training call with cfg->devices = n runs that call data-parallel over the n engines: the
// model / table / dataset handles the caller holds live on engine 0; replicas of the model (weights, Adam moments, step
// state, operand images) and of the embedding table on engines 1 .. n-1 are made by broadcast, the dataset is cut into
// per-rank shards (rank r owns rows [r, r+1) * B/n of every global batch of B rows), and n host threads -- one per engine --
// each run the ordinary per-rank data-parallel step loop (the one a one-process-per-GPU run executes) on their replica with
// the group's communicator switched on.  Replicas and shards are cached on the handles: a second call only re-broadcasts
// what changed in between (set_weights, set_rows, ...).
namespace {

// out[r][b * Bl + i][c] = in[(b * B + r * Bl + i)][c], `fill` where that row does not exist (4-byte elements)
__global__ __launch_bounds__(256) void shard_rows_kernel(const uint32_t* __restrict__ in, long long rows, int w, int B, int Bl, int W,
                                                         long long nb, uint32_t fill, uint32_t* __restrict__ out) {
  const long long per = nb * Bl * (long long)w, total = per * W;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long r = idx / per, rem = idx - r * per;
    const long long lr = rem / w; const int c = (int)(rem - lr * w);
    const long long b = lr / Bl; const int i = (int)(lr - b * Bl);
    const long long g = b * B + r * Bl + i;
    out[idx] = g < rows ? in[g * w + c] : fill;
  }
}

struct ShardPack {            // root-side staging of one array of the dataset: [W][nb * Bl][w]
  DevBuf<uint32_t> buf; size_t per = 0;
};

int pack_array(ShardPack& p, const void* in, long long rows, int w, int B, int W, long long nb, uint32_t fill) {
  const int Bl = B / W;
  p.per = (size_t)nb * Bl * w;
  if (!w || !in) { p.per = 0; return 0; }
  if (p.buf.alloc(p.per * W, false)) return -1;
  const long long total = (long long)p.per * W;
  const int cus = engine().compute_units > 0 ? engine().compute_units : 256;
  hipLaunchKernelGGL(shard_rows_kernel, dim3((unsigned)std::min<long long>(std::max<long long>(cdiv(total, 256), 1), 32 * cus)), dim3(256), 0,
                     engine().stream, static_cast<const uint32_t*>(in), rows, w, B, Bl, W, nb, fill, p.buf.p);
  GOCTR_HIP(hipGetLastError());
  return 0;
}

// collective: rank 0's pack -> every rank's `dst` (its per-rank slice)
int scatter_array(const ShardPack* root_pack, size_t per, void* dst) {
  Engine& e = engine();
  if (!per) return 0;
  const int W = e.world;
  std::vector<size_t> so((size_t)W, 0), sc((size_t)W, 0), ro((size_t)W, 0), rc((size_t)W, 0);
  if (e.rank == 0) for (int p = 0; p < W; ++p) { so[p] = (size_t)p * per; sc[p] = per; }
  rc[0] = per;
  return comm_alltoallv(e.rank == 0 ? (const void*)root_pack->buf.p : (const void*)dst, so.data(), sc.data(), dst, ro.data(), rc.data(), 4);
}

// collective: rank 0's model state -> this rank's replica
int model_broadcast(goctr_model* mk, int stp_root, float emb_lr_root) {
  Engine& e = engine();
  auto bc = [&](void* p, size_t bytes) -> int { return (p && bytes) ? comm_broadcast(p, bytes, 0) : 0; };
  if (bc(mk->W.p, sizeof(float) * mk->nflat) || bc(mk->Mo.p, sizeof(float) * mk->nflat) || bc(mk->Vo.p, sizeof(float) * mk->nflat) ||
      bc(mk->W1T.p, sizeof(float) * mk->W1T.n) || bc(mk->W2T.p, sizeof(float) * mk->W2T.n) || bc(mk->W0sT.p, sizeof(float) * mk->W0sT.n) ||
      bc(mk->Wimg.p, sizeof(float) * mk->Wimg.n) || bc(mk->Wx3.p, mk->x3_nch0 ? sizeof(unsigned short) * mk->Wx3.n : 0) ||
      bc(mk->st.p, sizeof(StepState) * 2)) return -1;
  if (e.rank != 0) {
    mk->stp = stp_root;
    if (mk->emb_lr != emb_lr_root) { mk->emb_lr = emb_lr_root; mk->graph.destroy(); }
    mk->w0pv_live = false;              // (rebuilt from the broadcast W0 by ensure_w0pv)
    mk->carry.valid = false;
    if (mk->ra_flag.p) GOCTR_HIP(hipMemsetAsync(mk->ra_flag.p, 0, sizeof(unsigned int), e.stream));
  }
  return 0;
}

struct CommCallScope {       // the group's communicator takes part in this call only
  Engine& e; bool prev;
  explicit CommCallScope(Engine& en) : e(en), prev(en.comm_enabled) { e.comm_enabled = true; }
  ~CommCallScope() { e.comm_enabled = prev; }
};

// Does this training call take the multi-device entry?  devices = n > 1; or devices = 1 on a ONE-engine group that
// goctr_init_devices built a communicator for (GOCTR_FORCE_COMM=1): the same entry with one rank -- ncclCommInitAll, the
// broadcast, the scatter and the split step with a one-rank RCCL communicator, which is all of mode 2 that a one-GPU box can run
bool multi_call(const goctr_model* m, const goctr_train_cfg* cfg) {
  if (cfg->devices > 1) return true;
  const Engine* e = m->eng;
  return cfg->devices == 1 && engine_count() == 1 && e->index == 0 && (e->nccl_comm || e->loop) && !e->comm_enabled;
}

// per_rank(model, table, shard, local cfg, rank) is the ordinary per-rank call
template <class Fn>
int train_multi(goctr_model* m, goctr_emb* emb, goctr_dataset* d, const goctr_train_cfg* cfg, Fn per_rank) {
  const int N = cfg->devices, B = cfg->batch;
  Engine* e0 = engine_at(0);
  GOCTR_CHECK(N == engine_count() && e0 && e0->world == N && (e0->loop || e0->nccl_comm),
              "cfg.devices = %d, but goctr_init_devices set up %d engine(s)", N, (e0 && (e0->loop || e0->nccl_comm)) ? e0->world : 1);
  GOCTR_CHECK(m->eng == e0 && (!emb || emb->eng == e0) && d->eng == e0, "multi-device training: the handles must live on engine 0");
  GOCTR_CHECK(B % N == 0, "multi-device training: batch %d is not a multiple of devices %d", B, N);
  if (comm_group_reset()) return -1;
  const int Bl = B / N;
  const long long nb = cdiv(d->rows, B);
  // ---- handles on the other engines (no collectives yet)
  bool new_model = false, new_emb = false;
  if ((int)m->reps.size() != N) { for (auto* r : m->reps) goctr_model_destroy(r); m->reps.assign((size_t)N, nullptr); }
  if (emb && (int)emb->reps.size() != N) { for (auto* r : emb->reps) goctr_emb_destroy(r); emb->reps.assign((size_t)N, nullptr); }
  const bool need_shard = (int)d->shards.size() != N || d->shard_B != B;
  if (need_shard) { for (auto* s : d->shards) goctr_dataset_destroy(s); d->shards.assign((size_t)N, nullptr); d->shard_B = 0; }
  for (int k = 0; k < N; ++k) {
    Engine* ek = engine_at(k);
    EngineScope on(ek);
    std::lock_guard<std::recursive_mutex> lk(ek->mu);
    if (k > 0 && !m->reps[k]) { if (goctr_model_create(&m->cfg, &m->reps[k])) return -1; new_model = true; }
    if (k > 0 && emb && !emb->reps[k]) { if (goctr_emb_create(emb->V, emb->D, nullptr, &emb->reps[k])) return -1; new_emb = true; }
    if (need_shard) {
      std::unique_ptr<goctr_dataset> s(new goctr_dataset);
      s->id_mode = d->id_mode; s->rows = nb * Bl; s->has_y = d->has_y;
      s->xcols = d->xcols; memcpy(s->ranges, d->ranges, sizeof s->ranges); s->U = d->U; s->C = d->C; s->T = d->T;
      const size_t R = (size_t)s->rows;
      if (d->id_mode) {
        if (s->ub_ids.alloc(R * d->T, false) || s->item_ids.alloc(R, false) || s->ufeat.alloc(R * d->U, false) || s->cfeat.alloc(R * d->C, false)) return -1;
      } else if (s->X.alloc(R * d->xcols, false)) return -1;
      if (d->has_y && s->Y.alloc(R, false)) return -1;
      GOCTR_HIP(hipStreamSynchronize(ek->stream));
      d->shards[k] = s.release();
    }
  }
  const bool model_sync = new_model || m->reps_gen + 1 != m->gen;
  const bool emb_sync = emb && (new_emb || emb->reps_version != emb->version);
  // ---- root-side staging of the shards
  ShardPack pX, pY, pub, pit, puf, pcf;
  const bool from_host = d->host_X != nullptr;
  if (need_shard && !from_host) {
    if (d->id_mode) {
      if (pack_array(pub, d->ub_ids.p, d->rows, d->T, B, N, nb, 0xFFFFFFFFu) || pack_array(pit, d->item_ids.p, d->rows, 1, B, N, nb, 0xFFFFFFFFu) ||
          pack_array(puf, d->ufeat.p, d->rows, d->U, B, N, nb, 0u) || pack_array(pcf, d->cfeat.p, d->rows, d->C, B, N, nb, 0u)) return -1;
    } else if (pack_array(pX, d->X.p, d->rows, d->xcols, B, N, nb, 0u)) return -1;
    if (d->has_y && pack_array(pY, d->Y.p, d->rows, 1, B, N, nb, 0u)) return -1;
  }
  goctr_train_cfg lcfg = *cfg;
  lcfg.batch = Bl; lcfg.devices = 1;
  const int stp_root = m->stp; const float emb_lr_root = m->emb_lr;
  const int rc = run_on_engines(N, [&](int k) -> int {
    Engine& e = engine();
    std::lock_guard<std::recursive_mutex> elk(e.mu);
    CommCallScope comm_on(e);
    // (once per communicator; every rank is here.  < 0: the probe lost the communicator -- fail the call on this rank, the
    // others see the abort in their next wait)
    if (comm_capturable() && env_int("GOCTR_DP_CAPTURE_COMM", 1) == 1 && comm_capture_selftest() < 0) return -1;
    goctr_model* mk = k == 0 ? m : m->reps[k];
    goctr_emb* ek = !emb ? nullptr : (k == 0 ? emb : emb->reps[k]);
    goctr_dataset* dk = d->shards[k];
    std::unique_lock<std::shared_mutex> lk(mk->mu, std::defer_lock);
    if (k > 0) { lk.lock(); ++mk->gen; }          // (rank 0: the caller holds its model's lock)
    int r = 0;
    if (model_sync) r = model_broadcast(mk, stp_root, emb_lr_root);
    if (!r && emb_sync) { r = comm_broadcast(ek->rows.p, sizeof(float) * (size_t)(emb->V + 1) * emb->D, 0); if (k > 0) ++ek->version; }
    if (!r && need_shard && from_host) {
      // this rank's rows of global batch b are host rows [b B + k Bl, b B + (k + 1) Bl), clipped at the dataset's end; what is
      // missing of a short last batch is zero rows (model.go:357-371 FillTensorRows pads it to the batch size)
      hipStream_t st = e.stream;
      auto rows_from_host = [&](float* dst, const float* src, int cols) -> int {
        for (long long b = 0; b < nb; ++b) {
          const long long g0 = b * B + (long long)k * Bl;
          const long long have = std::max<long long>(0, std::min<long long>(Bl, d->rows - g0));
          float* to = dst + (size_t)b * Bl * cols;
          if (have > 0) GOCTR_HIP(hipMemcpyAsync(to, src + (size_t)g0 * cols, sizeof(float) * (size_t)have * cols, hipMemcpyHostToDevice, st));
          if (have < Bl) GOCTR_HIP(hipMemsetAsync(to + (size_t)have * cols, 0, sizeof(float) * (size_t)(Bl - have) * cols, st));
        }
        return 0;
      };
      r = rows_from_host(dk->X.p, d->host_X, d->xcols);
      if (!r && d->has_y) r = rows_from_host(dk->Y.p, d->host_Y, 1);
      if (!r) { const hipError_t he = hipStreamSynchronize(st); if (he != hipSuccess) { set_error("per-rank upload: %s", hipGetErrorString(he)); r = -1; } }
    } else if (!r && need_shard) {
      if (d->id_mode) r = scatter_array(&pub, pub.per, dk->ub_ids.p) || scatter_array(&pit, pit.per, dk->item_ids.p) ||
                          scatter_array(&puf, puf.per, dk->ufeat.p) || scatter_array(&pcf, pcf.per, dk->cfeat.p);
      else r = scatter_array(&pX, pX.per, dk->X.p);
      if (!r && d->has_y) r = scatter_array(&pY, pY.per, dk->Y.p);
      if (!r && k == 0) r = hipStreamSynchronize(e.stream) == hipSuccess ? 0 : -1;     // (the staging buffers are released after the call)
    }
    // (goctr_engine_call_ms: this rank's own span of the call on its own stream -- bench.py --single-process reports it per rank)
    if (!e.call_begin) { (void)hipEventCreate(&e.call_begin); (void)hipEventCreate(&e.call_end); }
    e.call_timed = false;
    if (!r && e.call_begin) (void)hipEventRecord(e.call_begin, e.stream);
    if (!r) r = per_rank(mk, ek, dk, &lcfg, k);
    if (!r && e.call_end) e.call_timed = hipEventRecord(e.call_end, e.stream) == hipSuccess;
    if (r) {
      const std::string msg = goctr_last_error();
      comm_abort_on_failure();
      set_error("%s", msg.c_str());
    }
    return r;
  });
  if (rc) { m->reps_gen = ~0ull; if (emb) emb->reps_version = ~0ull; return -1; }
  if (need_shard) d->shard_B = B;
  m->reps_gen = m->gen;
  if (emb) emb->reps_version = emb->version;
  return 0;
}

}  // namespace

static int train_steps_locked(goctr_model* m, goctr_emb* emb, goctr_dataset* d, const goctr_train_cfg* cfg, int64_t first_batch, int n_steps,
                              float* costs);
static int train_dataset_locked(goctr_model* m, goctr_emb* emb, goctr_dataset* d, const goctr_train_cfg* cfg, float* epoch_costs,
                                int* epochs_run);

extern "C" {

// ------------------------------------------------------------------ training
int goctr_train_steps(goctr_model* m, goctr_emb* emb, goctr_dataset* d, const goctr_train_cfg* cfg,
                      int64_t first_batch, int n_steps, float* costs) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m && d && cfg && cfg->batch > 0 && n_steps >= 0, "goctr_train_steps: bad arguments");
  GOCTR_CHECK(d->has_y, "goctr_train_steps: dataset has no labels");
  GOCTR_CHECK(cfg->dropout_mode == 0 || cfg->dropout_mode == 2, "multi-step training supports dropout_mode 0 or 2");
  GOCTR_CHECK(n_steps <= COST_RING, "n_steps > %d per call", COST_RING);
  GOCTR_SAME_ENGINE(m, d); GOCTR_SAME_ENGINE(m, emb);
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  if (check_dataset(m, d, emb)) return -1;
  if (multi_call(m, cfg))
    return train_multi(m, emb, d, cfg, [&](goctr_model* mk, goctr_emb* ek, goctr_dataset* dk, const goctr_train_cfg* lc, int rank) {
      return train_steps_locked(mk, ek, dk, lc, first_batch, n_steps, rank == 0 ? costs : nullptr);
    });
  return train_steps_locked(m, emb, d, cfg, first_batch, n_steps, costs);
}

}  // extern "C"

static int train_steps_locked(goctr_model* m, goctr_emb* emb, goctr_dataset* d, const goctr_train_cfg* cfg, int64_t first_batch, int n_steps,
                              float* costs) {
  std::unique_lock<std::shared_mutex> rows_lk;          // embedding training writes the table: serving passes wait (lock order: model, table)
  if (m->emb_lr > 0.f && emb) rows_lk = std::unique_lock<std::shared_mutex>(emb->mu);
  const long long nb = cdiv(d->rows, cfg->batch);
  if (retarget_state(m, first_batch % nb, nb)) return -1;      // no host synchronisation on this path
  m->pend_no_costs = costs == nullptr;
  const int rs = run_steps(m, emb, d, cfg, n_steps);
  m->pend_no_costs = false;
  if (rs) {
    if (engine().comm_active()) {     // (keep this rank's error text; make the peers fail too instead of waiting in a collective)
      const std::string msg = goctr_last_error();
      comm_abort_on_failure();
      set_error("%s [data-parallel step failed on this rank: communicator aborted]", msg.c_str());
    }
    return -1;
  }
  if (costs) {
    if (comm_watch_stream()) return -1;
    if (m->costs.download(costs, n_steps)) return -1;
  }
  return 0;
}

extern "C" {

int goctr_model_replica(goctr_model* m, int rank, goctr_model** out) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m && out && rank >= 0, "goctr_model_replica: bad arguments");
  std::unique_lock<std::shared_mutex> lk(m->mu);
  *out = rank == 0 ? m : (rank < (int)m->reps.size() ? m->reps[rank] : nullptr);
  return 0;
}
int goctr_emb_replica(goctr_emb* e, int rank, goctr_emb** out) {
  GOCTR_ENTER_H(e);
  GOCTR_CHECK(e && out && rank >= 0, "goctr_emb_replica: bad arguments");
  *out = rank == 0 ? e : (rank < (int)e->reps.size() ? e->reps[rank] : nullptr);
  return 0;
}

int goctr_train_dataset(goctr_model* m, goctr_emb* emb, goctr_dataset* d, const goctr_train_cfg* cfg,
                        float* epoch_costs, int* epochs_run) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m && d && cfg && cfg->batch > 0 && cfg->epochs >= 0, "goctr_train_dataset: bad arguments");
  GOCTR_CHECK(d->has_y, "goctr_train_dataset: dataset has no labels");
  GOCTR_CHECK(cfg->dropout_mode == 0 || cfg->dropout_mode == 2, "multi-step training supports dropout_mode 0 or 2");
  GOCTR_SAME_ENGINE(m, d); GOCTR_SAME_ENGINE(m, emb);
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  if (check_dataset(m, d, emb)) return -1;
  if (multi_call(m, cfg))
    return train_multi(m, emb, d, cfg, [&](goctr_model* mk, goctr_emb* ek, goctr_dataset* dk, const goctr_train_cfg* lc, int rank) {
      int ran = 0;
      const int r = train_dataset_locked(mk, ek, dk, lc, rank == 0 ? epoch_costs : nullptr, &ran);
      if (rank == 0 && epochs_run) *epochs_run = ran;
      return r;
    });
  return train_dataset_locked(m, emb, d, cfg, epoch_costs, epochs_run);
}

}  // extern "C"

static int train_dataset_locked(goctr_model* m, goctr_emb* emb, goctr_dataset* d, const goctr_train_cfg* cfg, float* epoch_costs,
                                int* epochs_run) {
  std::unique_lock<std::shared_mutex> rows_lk;          // embedding training writes the table: serving passes wait (lock order: model, table)
  if (m->emb_lr > 0.f && emb) rows_lk = std::unique_lock<std::shared_mutex>(emb->mu);
  // a fresh solver per model.Train call (model.go:88)
  GOCTR_HIP(hipMemsetAsync(m->Mo.p, 0, sizeof(float) * m->nflat, engine().stream));
  GOCTR_HIP(hipMemsetAsync(m->Vo.p, 0, sizeof(float) * m->nflat, engine().stream));
  const long long nb = cdiv(d->rows, cfg->batch);
  if (engine().comm_active()) {
    // every rank issues one all-reduce per batch: unequal shard sizes would leave the shorter ranks' peers hanging
    double v[2] = {(double)nb, (double)nb * (double)nb};
    if (goctr_comm_allreduce_f64(v, 2)) return -1;
    const double w = (double)engine().eff_world();
    GOCTR_CHECK(v[0] == w * (double)nb && v[1] == w * (double)nb * (double)nb,
                "goctr_train_dataset: the ranks' shards have different batch counts (this rank: %lld batches of %d); "
                "shard the rows so that every rank steps the same number of times", nb, cfg->batch);
  }
  if (set_state(m, 0, 0, 0, nb)) return -1;
  float best = 3.402823466e+38f;  // math.MaxFloat32 (model.go:103)
  int no_improve = 0, e = 0;
  for (e = 0; e < cfg->epochs; ++e) {
    long long done = 0;
    unsigned slot0 = 0;
    while (done < nb) {  // keep each burst inside the cost ring
      const int burst = (int)std::min<long long>(nb - done, COST_RING / 2);
      if (run_steps(m, emb, d, cfg, burst)) {
        if (engine().comm_active()) {
          const std::string msg = goctr_last_error();
          comm_abort_on_failure();
          set_error("%s [data-parallel step failed on this rank: communicator aborted]", msg.c_str());
        }
        return -1;
      }
      done += burst;
    }
    (void)slot0;
    StepState s;
    if (get_state(m, &s)) return -1;
    float cost = 0.f;
    if (m->costs.download(&cost, 1, (s.slot - 1u) % COST_RING)) return -1;  // cost of the LAST batch (model.go:198)
    if (epoch_costs) epoch_costs[e] = cost;
    if (cost < best) { best = cost; no_improve = 0; } else no_improve++;
    if (cfg->early_stop != 0 && no_improve >= cfg->early_stop) { e++; break; }
  }
  if (epochs_run) *epochs_run = e;
  return 0;
}

extern "C" {

int goctr_train_dense(goctr_model* m, const float* X, const float* Y, int64_t rows, int xcols, const int ranges[8],
                      const goctr_train_cfg* cfg, float* epoch_costs, int* epochs_run) {
  GOCTR_ENTER_H(m);
  goctr_dataset* d = nullptr;
  GOCTR_CHECK(m && cfg, "goctr_train_dense: null argument");
  GOCTR_CHECK(Y != nullptr, "goctr_train_dense: labels required");
  if (cfg->devices > 1) {
    // n devices: no copy of X on engine 0 -- the ranks fetch their own rows from the caller's memory (goctr_dataset::host_X)
    GOCTR_CHECK(X && rows > 0 && xcols > 0 && ranges, "goctr_train_dense: bad arguments");
    d = new goctr_dataset;
    d->id_mode = false; d->rows = rows; d->xcols = xcols; d->has_y = true;
    memcpy(d->ranges, ranges, sizeof d->ranges);
    d->host_X = X; d->host_Y = Y;
  } else if (goctr_dataset_create_dense(X, Y, rows, xcols, ranges, &d)) return -1;
  int rc = goctr_train_dataset(m, nullptr, d, cfg, epoch_costs, epochs_run);
  if (!rc) rc = goctr_sync();
  {
    std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
    m->graph.destroy();  // (keyed on the dataset's generation id, so it could never be replayed again anyway)
  }
  goctr_dataset_destroy(d);
  return rc;
}

int goctr_loss_grad_dense(goctr_model* m, const float* X, const float* Y, int valid, int B, int xcols,
                          const int ranges[8], const goctr_train_cfg* cfg, uint32_t step, const float* m0,
                          const float* m1, float* cost, float* gW0, float* gW1, float* gW2, float* gatt0, float* y_out) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m && X && Y && cfg && valid > 0 && valid <= B, "goctr_loss_grad_dense: bad arguments");
  goctr_dataset* d = nullptr;
  if (goctr_dataset_create_dense(X, Y, valid, xcols, ranges, &d)) return -1;
  std::unique_ptr<goctr_dataset> guard(d);
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  if (check_dataset(m, d, nullptr)) return -1;
  if (ensure_workspace(m, B)) return -1;
  StepOpts o = opts_from(cfg);
  o.update = false;
  if (o.drop_mode == 1) {
    GOCTR_CHECK(m0 && m1, "dropout_mode 1 needs explicit masks");
    if (m->mask0.alloc((size_t)B * m->cfg.H1, false) || m->mask0.upload(m0, (size_t)B * m->cfg.H1)) return -1;
    if (m->mask1.alloc((size_t)B * m->cfg.H2, false) || m->mask1.upload(m1, (size_t)B * m->cfg.H2)) return -1;
  }
  StepState saved;
  if (get_state(m, &saved)) return -1;
  struct Restore {   // the caller's step counter / dropout stream position survives every exit path
    goctr_model* m; const StepState& s; bool armed = true;
    ~Restore() { if (armed) (void)set_state(m, s.gstep, s.slot, s.batch_idx, s.n_batches); }
  } restore{m, saved};
  if (set_state(m, step, 0, 0, 1)) return -1;
  RowSource src = make_source(d, nullptr);
  int rc = launch_forward(m, src, B, o) || launch_backward(m, src, B, o, false);
  if (rc) return -1;
  GOCTR_HIP(hipStreamSynchronize(engine().stream));
  if (gW0 && download_padded(m, m->G.p, GOCTR_W0, gW0, (size_t)m->I * m->cfg.H1)) return -1;
  if (gW1 && download_padded(m, m->G.p, GOCTR_W1, gW1, (size_t)m->cfg.H1 * m->cfg.H2)) return -1;
  if (gW2 && download_padded(m, m->G.p, GOCTR_W2, gW2, (size_t)m->cfg.H2)) return -1;
  if (gatt0) {
    if (m->cfg.kind == GOCTR_DIN) { if (download_padded(m, m->G.p, GOCTR_ATT0, gatt0, (size_t)m->cfg.T)) return -1; }
    else memset(gatt0, 0, sizeof(float) * m->cfg.T);
  }
  if (cost) {
    float s = 0.f;
    if (m->G.download(&s, 1, m->nflat)) return -1;
    *cost = -(s / (float)(B * engine().eff_world()));
  }
  if (y_out && m->yhat.download(y_out, B)) return -1;
  restore.armed = false;
  return set_state(m, saved.gstep, saved.slot, saved.batch_idx, saved.n_batches);
}

// ------------------------------------------------------------------ predict
static int predict_batches(goctr_model* m, goctr_emb* emb, goctr_dataset* d, int batch, int64_t first_batch,
                           int64_t n_batches, float* y_host) {
  if (check_dataset(m, d, emb)) return -1;
  // Rows are scored independently of their batch, so G consecutive batches can share launches (one gather and one forward
  // chain over G * batch rows): fewer and fuller launches.  PredBatchSize keeps its meaning at the boundary -- which rows a
  // call covers and how the short last batch is padded (model.go:337-347).  The scores agree with one-batch launches to
  // float32 rounding, not bit for bit: 16 384 rows take the 32-row-tile forward kernel, 4096 rows the 16-row-tile one
  // (322 instead of 418 M rows/s if the latter scored the groups too), and the two add the partial products of layer 1 in
  // different orders (tests/test_gpu_ctr.py bounds the difference; both are inside the 1e-5 parity bar vs the oracle).
  // Measured at DIN cfg3, PredBatchSize 4096: 250 / 351 / 416 / 444 M rows/s at G = 1 / 2 / 4 / 8 (GOCTR_PRED_GROUP) with one
  // workgroup per 32-row tile; since the forward-only kernel walks its tiles as one persistent workgroup per CU (round 3,
  // ctr_chain_x3.h: a tile's start hides behind its predecessor's tail) 516 / 573 M at G = 4 / 8 -- default 8.
  int G = std::max(1, env_int("GOCTR_PRED_GROUP", 8));
  while (G > 1 && (long long)batch * G > 32768) G /= 2;     // (a launch of 32 768 rows fills the chip; the workspace grows with G)
  // forward-only workspace of its own (h0, gates, yhat): the training workspace -- sized for the training batch, with its
  // slab buffers and captured step graphs -- is left alone
  if (m->pws.ensure(batch * G, m->Ip, m->cfg.T, m->H1p, m->H2p, !chain_ok(m), engine().stream)) return -1;
  const FwdBufs fb = m->pws.bufs();
  RowSource src = make_source(d, emb);
  StepOpts o;
  o.train = false;
  const long long nb = cdiv(d->rows, batch);
  // per-batch states are written up front so that no host stack memory is read asynchronously
  const int64_t CH = 4096;
  if (m->pst.ensure((size_t)std::min<int64_t>(n_batches, CH), false)) return -1;
  if (y_host && m->yall.ensure((size_t)d->rows, false)) return -1;
  std::vector<StepState> hs;
  for (int64_t k0 = 0; k0 < n_batches; k0 += CH) {
    const int64_t cnt = std::min<int64_t>(CH, n_batches - k0);
    hs.resize(cnt);
    std::vector<int> grp((size_t)cnt, 1);
    for (int64_t k = 0; k < cnt;) {
      const long long b = (first_batch + k0 + k) % nb;
      // a group: g whole batches that start at a multiple of g and do not run past the call or the dataset's last batch;
      // g = G, or the largest G / 2^j that still fits (the tail of a dataset keeps to the large-launch kernel as long as
      // two batches are left)
      int g = 1;
      for (int c = G; c > 1; c /= 2)
        if (b % c == 0 && k + c <= cnt && b + c <= nb) { g = c; break; }
      hs[k] = StepState{0u, 0u, b / g, nb};
      grp[k] = g;
      for (int j = 1; j < g; ++j) { hs[k + j] = hs[k]; grp[k + j] = 0; }
      k += g;
    }
    if (m->pst.upload(hs.data(), (size_t)cnt)) return -1;
    for (int64_t k = 0; k < cnt; ++k) {
      if (grp[k] == 0) continue;                      // (covered by the group that started before it)
      const int Bk = batch * grp[k];
      if (launch_forward(m, src, Bk, o, m->pst.p + k, &fb)) return -1;
      if (y_host) {
        const long long b = hs[k].batch_idx;
        const long long start = b * Bk, end = std::min<long long>(start + Bk, d->rows);
        // first end-start outputs (model.go:344-347), collected on the device: one copy to the host per call
        GOCTR_HIP(hipMemcpyAsync(m->yall.p + start, fb.yhat, sizeof(float) * (size_t)(end - start), hipMemcpyDeviceToDevice,
                                 engine().stream));
      }
    }
    if (k0 + CH < n_batches) GOCTR_HIP(hipStreamSynchronize(engine().stream));  // before the states are overwritten
  }
  if (y_host) {
    // (callers always score from batch 0: every row of [0, min(rows, n_batches * batch)) was written above)
    const long long n = std::min<long long>(d->rows, n_batches * (long long)batch);
    if (m->yall.download(y_host, (size_t)n)) return -1;
  }
  return 0;
}

int goctr_predict_dataset(goctr_model* m, goctr_emb* emb, goctr_dataset* d, int batch, float* y_out) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m && d && y_out && batch > 0, "goctr_predict_dataset: bad arguments");
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  return predict_batches(m, emb, d, batch, 0, cdiv(d->rows, batch), y_out);
}

int goctr_predict_steps(goctr_model* m, goctr_emb* emb, goctr_dataset* d, int batch, int64_t first_batch,
                        int n_batches) {
  GOCTR_ENTER_H(m);
  GOCTR_CHECK(m && d && batch > 0 && n_batches >= 0, "goctr_predict_steps: bad arguments");
  std::unique_lock<std::shared_mutex> lk(m->mu); ++m->gen;
  return predict_batches(m, emb, d, batch, first_batch, n_batches, nullptr);
}

}  // extern "C"

// ------------------------------------------------------------------ serving: recommend.BatchPredict / Rank / Predict
// recommend/rcmd.go:248-337: sample keys -> GetSampleVector rows -> PredictAbstract.Predict -> scores, called from concurrent
// gin handler goroutines (recommend/api.go:106-131: one user, a short itemIdList per request).  Everything GetSampleVector
// reads per key (rcmd.go:462-536) is resident in HBM -- the user / item feature tables (the contents of UserFeatureCache /
// ItemFeatureCache), the behaviour cache, the item-embedding table -- so one call is: keys (16 B each) to the device, one
// assembly launch, the forward launches, scores back.
//
// Concurrency.  These entry points do not take the engine lock and do not use the engine's main stream.  A call borrows a
// SERVING SLOT: its own HIP stream, pinned host staging for keys and scores (one H2D and one D2H copy per pass, both
// asynchronous on the slot's stream; no per-call allocation, no std::vector copies), the id-mode rows assembled from the
// keys and a forward workspace.  It holds the model's lock SHARED (training holds it exclusive), and its stream waits for
// the event the last weight-writing call recorded on the main stream -- training is asynchronous.  Slots: GOCTR_SERVE_SLOTS
// (default 8), created on first use; further callers wait for a free one, first come first served (ServePool).
//
// Micro-batching.  A Rank request is tens to hundreds of rows: three small launches and two copies whose cost is latency,
// not work.  Requests of <= GOCTR_SERVE_COALESCE rows (default 1024) go through a combining queue per recsys: the first
// arrival becomes the leader and serves its own request; whatever arrives on the same model while that pass is in flight is
// taken over as ONE pass (<= 4096 rows) by the next leader -- one of the waiting callers, so no thread serves others after
// its own result is ready.  Rows are scored independently and passes of < 8192 rows all run the same forward kernel
// (ctr_fwd16_kernel), so a request's scores are bit-identical whether or not, and with whatever, it was coalesced.
struct goctr_recsys {
  goctr::Engine* const eng = &goctr::engine();   // the engine (device, streams, arena) the handle was created on
  goctr_ubcache* ub = nullptr;    // not owned
  goctr_emb* emb = nullptr;       // not owned
  int64_t n_users = 0, n_items = 0; int U = 0, C = 0;
  DevBuf<float> user_table, item_table;
  // combining queue of small requests (micro-batcher)
  struct Req;
  std::mutex qmu; std::condition_variable qcv;
  std::vector<Req*> queue; std::atomic<bool> leader{false};
};

namespace {

constexpr int64_t SERVE_PASS_ROWS = 65536;      // rows one pass of a slot scores (larger requests: several passes)
constexpr int64_t SERVE_COALESCE_ROWS = 4096;   // rows one coalesced pass may hold (< 8192: always ctr_fwd16_kernel)

// one request's keys and outputs (host pointers of the caller)
struct KeySeg {
  const int32_t* users; int32_t user_all;       // users == null: every key has user_all (Rank)
  const int32_t* items;
  const int64_t* ts; int64_t ts_all;            // ts == null: every key has ts_all
  int64_t n;
  float* scores; uint8_t* failed; int64_t n_failed;
};

struct ServeSlot {
  hipStream_t stream = nullptr;
  int64_t cap = 0; int T = 0, U = 0, C = 0;
  // pinned staging: in = [ts i64 x N | users i32 x N | items i32 x N], out = [scores f32 x Br | failed u8 x N]
  char* h_in = nullptr; char* h_out = nullptr;
  unsigned* h_done = nullptr; unsigned epoch = 0;   // behind the failed flags in h_out: one word per 16-row workgroup (serve_keys_pass)
  // the keys of a zero-copy pass in fine-grained DEVICE memory that the host stores into over the PCIe BAR (large-BAR systems): the
  // kernel's first loads are local instead of a PCIe read round trip.  Null: the kernels read the pinned h_in.
  char* in_bar = nullptr; std::vector<void*> retired_dev;
  std::vector<void*> retired;      // outgrown pinned buffers (see ensure_keys)
  DevBuf<char> d_in, d_out;
  DevBuf<int32_t> ub_ids, item_ids; DevBuf<float> ufeat, cfeat;
  FwdWs ws;
  DevBuf<StepState> st;            // one all-zero state: "batch 0 of 1"
  DevBuf<float> X; size_t capX = 0;   // dense rows (goctr_predict_dense)
  ~ServeSlot() {
    for (void* p : retired_dev) (void)hipFree(p);
    if (in_bar) (void)hipFree(in_bar);
    for (void* p : retired) (void)hipHostFree(p);
    if (h_in) (void)hipHostFree(h_in);
    if (h_out) (void)hipHostFree(h_out);
    if (stream) (void)hipStreamDestroy(stream);
  }
  int init() {
    GOCTR_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (st.alloc(1, false)) return -1;
    const StepState z{0u, 0u, 0, 1};
    GOCTR_HIP(hipMemcpyAsync(st.p, &z, sizeof z, hipMemcpyHostToDevice, stream));
    GOCTR_HIP(hipStreamSynchronize(stream));
    return 0;
  }
  // room for n keys of a model / recsys with these widths
  int ensure_keys(int64_t n, int Tn, int Un, int Cn) {
    if (n <= cap && Tn == T && Un == U && Cn == C) return 0;
    GOCTR_HIP(hipStreamSynchronize(stream));
    const int64_t want = std::max<int64_t>(std::max<int64_t>(n, 256), std::min<int64_t>(2 * cap, SERVE_PASS_ROWS));
    const size_t Br = (size_t)round_up((int)want, 32);
    cap = 0;                                   // (a failure below must not leave the old capacity next to missing buffers)
    // (outgrown pinned buffers are kept until the slot goes: hipHostFree waits for the whole device, which would invalidate the
    // stream capture of a thread that is building step graphs meanwhile -- a handful of geometric growths per slot at most)
    if (h_in) { retired.push_back(h_in); h_in = nullptr; }
    if (h_out) { retired.push_back(h_out); h_out = nullptr; }
    GOCTR_HIP(hipHostMalloc((void**)&h_in, (size_t)want * 16, hipHostMallocDefault));
    if (in_bar) { retired_dev.push_back(in_bar); in_bar = nullptr; }
    if (env_int("GOCTR_SERVE_BAR", 1) != 0) in_bar = static_cast<char*>(bar_alloc((size_t)want * 16));   // (null: the pinned buffer serves)
    const size_t done_off = (Br * 4 + (size_t)want + 63) / 64 * 64, done_n = (size_t)want / 16 + 1;
    GOCTR_HIP(hipHostMalloc((void**)&h_out, done_off + 4 * done_n, hipHostMallocDefault));
    h_done = reinterpret_cast<unsigned*>(h_out + done_off);
    memset(h_done, 0, 4 * done_n); epoch = 0;
    if (d_in.alloc((size_t)want * 16, false) || d_out.alloc(Br * 4 + (size_t)want, false) ||
        ub_ids.alloc((size_t)want * Tn, false) || item_ids.alloc((size_t)want, false) ||
        ufeat.alloc((size_t)want * Un, false) || cfeat.alloc((size_t)want * Cn, false)) return -1;
    cap = want; T = Tn; U = Un; C = Cn;
    return 0;
  }
};

// Slots are handed out FAIRLY: a released slot goes straight to the longest-waiting caller (FIFO hand-off, no barging).  With a
// plain condition variable a caller in a closed loop re-took the slot it had just released before the woken waiter was
// scheduled, and waiters starved: 8 callers on 4 slots had a p99 of 300 - 870 us and a worst case of 50 ms against a p50 of
// 40 us (round 3's serving tail; profiles/r04_serve_tail.txt).  A waiter first spins on its hand-off word for about one pass
// (~50 us) -- a futex wake-up costs as much as the pass it waits for -- and only then blocks.
struct ServePool {
  std::mutex mu;
  std::vector<std::unique_ptr<ServeSlot>> all; std::vector<ServeSlot*> idle;
  struct Waiter { std::atomic<ServeSlot*> got{nullptr}; std::condition_variable cv; bool blocked = false; };
  std::deque<Waiter*> waiters;
  // try_only: null instead of waiting when every slot is busy (the micro-batcher's "is a slot free right now?")
  ServeSlot* acquire(bool try_only = false) {
    Waiter w;
    {
      std::unique_lock<std::mutex> lk(mu);
      const size_t max_slots = (size_t)std::max(1, env_int("GOCTR_SERVE_SLOTS", 8));
      if (waiters.empty()) {
        if (!idle.empty()) { ServeSlot* s = idle.back(); idle.pop_back(); return s; }
        if (all.size() < max_slots) {
          std::unique_ptr<ServeSlot> s(new ServeSlot);
          if (s->init()) return nullptr;
          all.push_back(std::move(s));
          return all.back().get();
        }
      }
      if (try_only) return nullptr;
      waiters.push_back(&w);
    }
    for (int spin = 0; spin < 20000; ++spin) {            // ~50 us
      if (ServeSlot* s = w.got.load(std::memory_order_acquire)) return s;
      __builtin_ia32_pause();
    }
    std::unique_lock<std::mutex> lk(mu);
    w.blocked = true;
    w.cv.wait(lk, [&] { return w.got.load(std::memory_order_acquire) != nullptr; });
    return w.got.load(std::memory_order_acquire);
  }
  void release(ServeSlot* s) {
    std::lock_guard<std::mutex> lk(mu);
    if (waiters.empty()) { idle.push_back(s); return; }
    Waiter* w = waiters.front();
    waiters.pop_front();
    // (w lives on the waiter's stack.  A SPINNING waiter returns the moment it sees `got`: nothing of w may be touched after
    // the store.  A BLOCKED waiter cannot return before it re-takes `mu`, which we hold until after the notify.)
    const bool blocked = w->blocked;
    w->got.store(s, std::memory_order_release);
    if (blocked) w->cv.notify_one();
  }
  // (goctr_*_destroy of something a slot may have buffers sized for: nothing to do -- slots hold no handle pointers)
};
// (never destroyed: a static destructor would release streams and pinned buffers after the HIP runtime has shut down)
ServePool& serve_pool() {
  Engine& e = engine();                        // (slots hold streams and buffers of this engine's device)
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (!e.serve_pool) e.serve_pool = new ServePool;
  return *static_cast<ServePool*>(e.serve_pool);
}
struct SlotLease {
  ServePool& pool;
  ServeSlot* s;
  SlotLease() : pool(serve_pool()), s(pool.acquire()) {}
  ~SlotLease() { if (s) pool.release(s); }
};

// A serving pass must see every weight write queued on the main stream so far (training is asynchronous).  The calling
// thread waits for the event on the HOST: a hipStreamWaitEvent from the slot's stream fails ("dependency created on
// uncaptured work in another stream") whenever another thread happens to be capturing a step graph on the main stream at
// that moment -- HIP judges the event by its stream's current capture state.  The caller holds the model's lock shared, so
// no new weight write can be queued while it waits; once the event has completed nothing is pending until the next one.
int serve_wait_weights(goctr_model* m, ServeSlot* s) {
  (void)s;
  if (m->weights_pending.load(std::memory_order_acquire) && m->ev_weights) {
    GOCTR_HIP(hipEventSynchronize(m->ev_weights));
    m->weights_pending.store(false, std::memory_order_release);
  }
  return 0;
}
// the same for the embedding rows a pass gathers (written by embedding training of ANY model that was given the table);
// the caller holds the table's lock shared
int serve_wait_rows(goctr_emb* e) {
  if (e && e->rows_pending.load(std::memory_order_acquire) && e->ev_rows) {
    GOCTR_HIP(hipEventSynchronize(e->ev_rows));
    e->rows_pending.store(false, std::memory_order_release);
  }
  return 0;
}

// One pass: the keys of `segs` (N rows in all, N <= SERVE_PASS_ROWS) -> scores / failed flags of every segment.
// Caller holds m->mu shared and owns the slot.
int serve_keys_pass(goctr_model* m, goctr_recsys* r, ServeSlot* s, KeySeg* const* segs, int nseg) {
  int64_t N = 0;
  for (int k = 0; k < nseg; ++k) N += segs[k]->n;
  const int T = m->cfg.T;
  // (sized for a full coalesced pass from the first call on: a slot that grew with every larger pass paid a pinned
  // re-allocation + a stream synchronisation each time -- part of round 3's serving tail)
  const int64_t cap_rows = std::max<int64_t>(N, SERVE_COALESCE_ROWS);
  if (s->ensure_keys(cap_rows, T, r->U, r->C)) return -1;
  if (s->ws.ensure((int)cap_rows, m->Ip, T, m->H1p, m->H2p, !chain_ok(m), s->stream)) return -1;
  const size_t Br = (size_t)round_up((int)N, 32);
  // Small passes read the keys and write the scores without copy commands on the stream (GOCTR_SERVE_ZEROCOPY=rows, default 4096;
  // 0 = never): a pass is one launch (ctr_serve16_kernel) and one wait.  The keys (16 B per row) are stored by the host straight into
  // device memory over the PCIe BAR where the system has a large BAR (GOCTR_SERVE_BAR=0: off), else the kernels read the pinned
  // host buffer; the scores and flags (5 B per row) are written to pinned host memory from inside the kernels.  Larger passes keep
  // the two DMA copies.
  const bool zc = N <= 4096;
  const bool bar = zc && s->in_bar != nullptr && env_int("GOCTR_SERVE_BAR", 1) != 0;
  char* const key_dst = bar ? s->in_bar : s->h_in;      // (written only, front to back: fine for a write-combined mapping)
  long long* hts = reinterpret_cast<long long*>(key_dst);
  int32_t* hus = reinterpret_cast<int32_t*>(key_dst + 8 * N);
  int32_t* hit = reinterpret_cast<int32_t*>(key_dst + 12 * N);
  int64_t o = 0;
  for (int k = 0; k < nseg; ++k) {
    const KeySeg& g = *segs[k];
    if (g.ts) memcpy(hts + o, g.ts, sizeof(int64_t) * (size_t)g.n);
    else for (int64_t i = 0; i < g.n; ++i) hts[o + i] = g.ts_all;
    if (g.users) memcpy(hus + o, g.users, sizeof(int32_t) * (size_t)g.n);
    else for (int64_t i = 0; i < g.n; ++i) hus[o + i] = g.user_all;
    memcpy(hit + o, g.items, sizeof(int32_t) * (size_t)g.n);
    o += g.n;
  }
  if (bar) __builtin_ia32_sfence();                   // the key stores are out before the launch's doorbell
  if (!zc) GOCTR_HIP(hipMemcpyAsync(s->d_in.p, s->h_in, (size_t)N * 16, hipMemcpyHostToDevice, s->stream));
  if (serve_wait_weights(m, s) || serve_wait_rows(r->emb)) return -1;
  const char* in_base = bar ? s->in_bar : (zc ? s->h_in : s->d_in.p);
  char* out_base = zc ? s->h_out : s->d_out.p;
  const long long* dts = reinterpret_cast<const long long*>(in_base);
  const int32_t* dus = reinterpret_cast<const int32_t*>(in_base + 8 * N);
  const int32_t* dit = reinterpret_cast<const int32_t*>(in_base + 12 * N);
  float* dscore = reinterpret_cast<float*>(out_base);
  unsigned char* dfail = reinterpret_cast<unsigned char*>(out_base + 4 * Br);
  const goctr_ubcache* c = r->ub;
  StreamScope on_slot(s->stream);
  RowSource src{};
  src.rows = N; src.id_mode = 1; src.emb = r->emb->rows.p; src.V = r->emb->V;
  // Embedding widths with a compile-time attention variant (D = 4 .. 64, a power of two) look the keys up INSIDE attn_fwd
  // (attn_fwd_keys_kernel, or the whole pass as ctr_serve16_kernel): the assembled rows (behaviour ids, feature rows) never
  // exist in HBM.  Other widths, tables of 4 GB and more, or GOCTR_SERVE_FUSE=0, assemble first.
  int fgroups = 0;
  const bool fuse = env_int("GOCTR_SERVE_FUSE", 1) != 0 && attn_fast_mode(m, src, &fgroups) != 0 && fgroups <= 16;   // (D = 4 .. 64, table < 4 GB)
  if (fuse) {
    src.k_users = dus; src.k_items = dit; src.k_ts = dts; src.k_failed = dfail;
    src.ub_off = c ? c->off.p : nullptr; src.ub_items = c ? c->items.p : nullptr; src.ub_ts = c ? c->ts.p : nullptr;
    src.user_table = r->user_table.p; src.item_table = r->item_table.p; src.n_users = r->n_users; src.n_items = r->n_items;
  } else {
    hipLaunchKernelGGL(assemble_keys_kernel, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, s->stream,
                       c ? c->off.p : (const long long*)nullptr, c ? c->items.p : (const int32_t*)nullptr,
                       c ? c->ts.p : (const long long*)nullptr, (long long)r->n_users, r->user_table.p, r->U, r->item_table.p,
                       (long long)r->n_items, r->C, dus, dit, dts, (long long)N, T, s->ub_ids.p, s->ufeat.p, s->cfeat.p,
                       s->item_ids.p, dfail);
    GOCTR_HIP(hipGetLastError());
    src.ub_ids = s->ub_ids.p; src.item_ids = s->item_ids.p; src.ufeat = s->ufeat.p; src.cfeat = s->cfeat.p;
  }
  FwdBufs fb = s->ws.bufs();
  fb.yhat = dscore;
  StepOpts op;
  op.train = false;
  // A zero-copy pass of one launch: the kernel's workgroups stamp this pass's number into the pinned buffer behind their scores,
  // and the host watches the stamps instead of waiting for the stream's completion signal -- for passes of up to
  // GOCTR_SERVE_POLL_ROWS rows (default 512; 0 = never): the release fence in front of a stamp writes back the rows' h0 from the
  // L2, which costs a 2048-row pass more than the wait saves (profiles/r05_serve_poll.txt; 256 rows until the keys went over the BAR).
  unsigned n_stamps = 0;
  if (fuse && serve16_ok(m, src, (int)N)) {          // key lookup + attention + forward chain: one launch
    const bool poll = zc && N <= (int64_t)env_int("GOCTR_SERVE_POLL_ROWS", 512);
    if (poll) { if (++s->epoch == 0) s->epoch = 1; n_stamps = (unsigned)cdiv(N, 16); }
    if (launch_serve16(m, src, (int)N, s->st.p, fb, poll ? s->h_done : nullptr, s->epoch)) return -1;
  } else
  if (launch_forward(m, src, (int)N, op, s->st.p, &fb)) return -1;
  bool want_failed = false;
  for (int k = 0; k < nseg; ++k) want_failed = want_failed || segs[k]->failed || segs[k]->n_failed >= 0;
  // scores and flags are adjacent: one copy back (the gap between them is < 128 bytes)
  const size_t out_bytes = want_failed ? 4 * Br + (size_t)N : 4 * (size_t)N;
  if (!zc) GOCTR_HIP(hipMemcpyAsync(s->h_out, s->d_out.p, out_bytes, hipMemcpyDeviceToHost, s->stream));
  bool stamped = false;
  if (n_stamps) {                                    // (2 ms without the stamps: the stream wait, which also reports a fault)
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned done = 0, spins = 0;;) {
      while (done < n_stamps && __atomic_load_n(s->h_done + done, __ATOMIC_ACQUIRE) == s->epoch) ++done;
      if (done == n_stamps) { stamped = true; break; }
      if ((++spins & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
      __builtin_ia32_pause();
    }
  }
  if (!stamped) GOCTR_HIP(hipStreamSynchronize(s->stream));
  const float* hs = reinterpret_cast<const float*>(s->h_out);
  const unsigned char* hf = reinterpret_cast<const unsigned char*>(s->h_out + 4 * Br);
  o = 0;
  for (int k = 0; k < nseg; ++k) {
    KeySeg& g = *segs[k];
    memcpy(g.scores, hs + o, sizeof(float) * (size_t)g.n);
    if (want_failed) {
      if (g.failed) memcpy(g.failed, hf + o, (size_t)g.n);
      int64_t cnt = 0;
      for (int64_t i = 0; i < g.n; ++i) cnt += hf[o + i] != 0;
      g.n_failed = cnt;
    }
    o += g.n;
  }
  return 0;
}

}  // namespace

struct goctr_recsys::Req {
  goctr_model* m; KeySeg seg; int rc = 0; std::atomic<bool> done{false}; std::string err;
  Req(goctr_model* mm, const KeySeg& s) : m(mm), seg(s) {}
};

namespace {

// a request of any size on a slot of its own (several passes when it exceeds SERVE_PASS_ROWS)
int serve_keys_direct(goctr_model* m, goctr_recsys* r, KeySeg& g) {
  SlotLease lease;
  if (!lease.s) return -1;
  const int64_t want_failed = g.n_failed;
  int64_t total_failed = 0;
  for (int64_t o = 0; o < g.n; o += SERVE_PASS_ROWS) {
    KeySeg part = g;
    part.n = std::min<int64_t>(SERVE_PASS_ROWS, g.n - o);
    if (g.users) part.users = g.users + o;
    part.items = g.items + o;
    if (g.ts) part.ts = g.ts + o;
    part.scores = g.scores + o;
    if (g.failed) part.failed = g.failed + o;
    part.n_failed = want_failed;
    KeySeg* one = &part;
    if (serve_keys_pass(m, r, lease.s, &one, 1)) return -1;
    if (part.n_failed > 0) total_failed += part.n_failed;
  }
  g.n_failed = total_failed;
  return 0;
}

// the micro-batcher (see the section comment)
int serve_keys_coalesced(goctr_model* m, goctr_recsys* r, KeySeg& g) {
  goctr_recsys::Req me(m, g);
  std::unique_lock<std::mutex> lk(r->qmu);
  r->queue.push_back(&me);
  while (!me.done) {
    if (r->leader.load(std::memory_order_acquire)) {
      // a pass is in flight: it ends within tens of microseconds -- spin for about that long before paying a futex sleep + wake
      lk.unlock();
      for (int spin = 0; spin < 30000; ++spin) {
        if (me.done.load(std::memory_order_acquire) || !r->leader.load(std::memory_order_acquire)) break;
        __builtin_ia32_pause();
      }
      lk.lock();
      if (!me.done.load(std::memory_order_acquire) && r->leader.load(std::memory_order_acquire)) r->qcv.wait(lk);
      continue;
    }
    // lead one pass: the longest prefix of the queue on one model that fits a pass (always contains the front)
    r->leader = true;
    std::vector<goctr_recsys::Req*> batch;
    int64_t rows = 0;
    goctr_model* bm = r->queue.front()->m;
    size_t take = 0;
    for (; take < r->queue.size(); ++take) {
      goctr_recsys::Req* q = r->queue[take];
      if (q->m != bm || (take > 0 && rows + q->seg.n > SERVE_COALESCE_ROWS)) break;
      rows += q->seg.n;
      batch.push_back(q);
    }
    r->queue.erase(r->queue.begin(), r->queue.begin() + (long)take);
    lk.unlock();
    int rc = 0;
    std::string err;
    {
      SlotLease lease;
      std::vector<KeySeg*> segs;
      for (auto* q : batch) segs.push_back(&q->seg);
      rc = lease.s ? serve_keys_pass(bm, r, lease.s, segs.data(), (int)segs.size()) : -1;
      if (rc) err = goctr_last_error();
    }
    lk.lock();
    for (auto* q : batch) { q->rc = rc; q->err = err; q->done.store(true, std::memory_order_release); }
    r->leader.store(false, std::memory_order_release);
    r->qcv.notify_all();
  }
  lk.unlock();
  if (me.rc) set_error("%s", me.err.c_str());
  g = me.seg;
  return me.rc;
}

int serve_keys(goctr_model* m, goctr_recsys* r, KeySeg& g, int64_t* n_failed) {
  std::shared_lock<std::shared_mutex> lm(m->mu);        // weights stay put while a slot reads them
  std::shared_lock<std::shared_mutex> le(r->emb->mu);   // ... and so do the embedding rows (lock order: model, table)
  const int64_t coalesce = std::min<int64_t>(std::max(0, env_int("GOCTR_SERVE_COALESCE", 1024)), SERVE_COALESCE_ROWS);
  // Small requests: straight onto a slot when one is free RIGHT NOW (nothing to wait for, nothing to combine with: coalescing
  // would only add the wait for the pass in flight -- it raised the 8-caller p50 at n = 256 from 38 to 63 us in round 3);
  // when every slot is busy they join the combining queue, whose next leader scores everything that queued up in ONE pass.
  int rc;
  if (g.n <= coalesce) {
    ServeSlot* free_slot = serve_pool().acquire(true);
    if (free_slot) {
      KeySeg* one = &g;
      const int64_t want_failed = g.n_failed;
      rc = serve_keys_pass(m, r, free_slot, &one, 1);
      serve_pool().release(free_slot);
      if (want_failed < 0) g.n_failed = -1;
    } else rc = serve_keys_coalesced(m, r, g);
  } else rc = serve_keys_direct(m, r, g);
  if (!rc && n_failed) *n_failed = g.n_failed;
  return rc;
}

}  // namespace

extern "C" {

int goctr_recsys_create(goctr_ubcache* c, goctr_emb* emb, const float* user_table, int64_t n_users, int U,
                        const float* item_table, int64_t n_items, int C, goctr_recsys** out) {
  GOCTR_ENTER_H(emb);
  GOCTR_SAME_ENGINE(c, emb);
  GOCTR_CHECK(emb && out && n_users > 0 && n_items > 0 && U >= 0 && C >= 0, "goctr_recsys_create: bad arguments");
  GOCTR_CHECK((U == 0 || user_table) && (C == 0 || item_table), "goctr_recsys_create: feature table missing");
  GOCTR_CHECK(!c || c->n_users == n_users, "goctr_recsys_create: user table has %lld rows, the behaviour cache %lld users",
              (long long)n_users, c ? (long long)c->n_users : 0LL);
  std::unique_ptr<goctr_recsys> r(new goctr_recsys);
  r->ub = c; r->emb = emb; r->n_users = n_users; r->n_items = n_items; r->U = U; r->C = C;
  if (r->user_table.alloc((size_t)n_users * U, false) || (U && r->user_table.upload(user_table, (size_t)n_users * U))) return -1;
  if (r->item_table.alloc((size_t)n_items * C, false) || (C && r->item_table.upload(item_table, (size_t)n_items * C))) return -1;
  *out = r.release();
  return 0;
}

void goctr_recsys_destroy(goctr_recsys* r) {
  if (!r) return;
  EngineScope on(r->eng);
  std::lock_guard<std::recursive_mutex> lk(r->eng->mu);
  // (serving passes are synchronous: none is in flight once its caller returned; no device-wide wait -- see goctr_model_destroy)
  if (engine().inited) { (void)hipStreamSynchronize(engine().stream); (void)hipStreamSynchronize(engine().side); }
  delete r;
}

int goctr_batch_predict(goctr_model* m, goctr_recsys* r, const int32_t* users, const int32_t* items, const int64_t* ts,
                        int64_t n, int batch, float* scores, uint8_t* failed, int64_t* n_failed) {
  EngineScope on(handle_engine(m));
  if (require_engine()) return -1;
  GOCTR_CHECK(m && r && users && items && scores && n >= 0 && batch > 0, "goctr_batch_predict: bad arguments");
  GOCTR_SAME_ENGINE(m, r);
  GOCTR_CHECK(r->emb->D == m->cfg.D && r->U == m->cfg.U && r->C == m->cfg.C, "goctr_batch_predict: recsys dims (U=%d,C=%d,D=%d) != model (U=%d,C=%d,D=%d)",
              r->U, r->C, r->emb->D, m->cfg.U, m->cfg.C, m->cfg.D);
  if (n_failed) *n_failed = 0;
  if (n == 0) return 0;
  // rcmd.go:293-296: a failing FIRST key aborts the call (there is no row width to build a zero row from yet)
  GOCTR_CHECK(users[0] >= 0 && users[0] < r->n_users && items[0] >= 0 && items[0] < r->n_items,
              "get sample vector error: first key (user %d, item %d) has no features", users[0], items[0]);
  // (PredBatchSize `batch` decides how model.Predict cuts the rows, model.go:337-347; a row's score does not depend on it)
  KeySeg g{users, 0, items, ts, 0, n, scores, failed, (failed || n_failed) ? 0 : -1};
  return serve_keys(m, r, g, n_failed);
}

int goctr_rank(goctr_model* m, goctr_recsys* r, int32_t user, const int32_t* items, int64_t n, int64_t ts, int batch,
               float* scores, uint8_t* failed, int64_t* n_failed) {
  EngineScope on(handle_engine(m));
  if (require_engine()) return -1;
  GOCTR_CHECK(m && r && items && scores && n >= 0 && batch > 0, "goctr_rank: bad arguments");
  GOCTR_SAME_ENGINE(m, r);
  GOCTR_CHECK(r->emb->D == m->cfg.D && r->U == m->cfg.U && r->C == m->cfg.C, "goctr_rank: recsys dims (U=%d,C=%d,D=%d) != model (U=%d,C=%d,D=%d)",
              r->U, r->C, r->emb->D, m->cfg.U, m->cfg.C, m->cfg.D);
  if (n_failed) *n_failed = 0;
  if (n == 0) return 0;
  GOCTR_CHECK(user >= 0 && user < r->n_users && items[0] >= 0 && items[0] < r->n_items,
              "get sample vector error: first key (user %d, item %d) has no features", user, items[0]);
  KeySeg g{nullptr, user, items, nullptr, ts, n, scores, failed, (failed || n_failed) ? 0 : -1};
  return serve_keys(m, r, g, n_failed);
}

// model.Predict's own convention (model/model.go:242-352): `rows` dense TrainSample rows in HOST memory -> y_out [rows].
// Concurrent like the two above (PredictAbstract.Predict is what the gin handlers end up in): a slot of its own, the rows
// travel in passes of <= 64 MB.
int goctr_predict_dense(goctr_model* m, const float* X, int64_t rows, int xcols, const int ranges[8], int batch,
                        float* y_out) {
  EngineScope on(handle_engine(m));
  if (require_engine()) return -1;
  GOCTR_CHECK(m && X && y_out && ranges && rows >= 0 && batch > 0 && xcols > 0, "goctr_predict_dense: bad arguments");
  if (rows == 0) return 0;
  goctr_dataset shape;                       // (only its ranges are looked at)
  shape.id_mode = false; shape.rows = rows; shape.xcols = xcols;
  memcpy(shape.ranges, ranges, sizeof shape.ranges);
  std::shared_lock<std::shared_mutex> lm(m->mu);
  if (check_dataset(m, &shape, nullptr)) return -1;
  SlotLease lease;
  ServeSlot* s = lease.s;
  if (!s) return -1;
  const int64_t pass = std::max<int64_t>(32, std::min<int64_t>(SERVE_PASS_ROWS, ((int64_t)64 << 20) / ((int64_t)xcols * 4) / 32 * 32));
  StreamScope on_slot(s->stream);
  for (int64_t o = 0; o < rows; o += pass) {
    const int64_t N = std::min(pass, rows - o);
    if (s->capX < (size_t)N * xcols) {
      GOCTR_HIP(hipStreamSynchronize(s->stream));
      if (s->X.alloc((size_t)std::min<int64_t>(pass, rows) * xcols, false)) return -1;
      s->capX = (size_t)std::min<int64_t>(pass, rows) * xcols;
    }
    if (s->ensure_keys(N, m->cfg.T, m->cfg.U, m->cfg.C)) return -1;       // (for its pinned score staging and d_out)
    if (s->ws.ensure((int)N, m->Ip, m->cfg.T, m->H1p, m->H2p, !chain_ok(m), s->stream)) return -1;
    GOCTR_HIP(hipMemcpyAsync(s->X.p, X + (size_t)o * xcols, (size_t)N * xcols * sizeof(float), hipMemcpyHostToDevice, s->stream));
    if (serve_wait_weights(m, s)) return -1;
    RowSource src{};
    src.rows = N; src.id_mode = 0; src.X = s->X.p; src.xcols = xcols;
    src.r_u = ranges[0]; src.r_ub = ranges[2]; src.r_v = ranges[4]; src.r_c = ranges[6];
    FwdBufs fb = s->ws.bufs();
    fb.yhat = reinterpret_cast<float*>(s->d_out.p);
    StepOpts op;
    op.train = false;
    if (launch_forward(m, src, (int)N, op, s->st.p, &fb)) return -1;
    GOCTR_HIP(hipMemcpyAsync(s->h_out, s->d_out.p, (size_t)N * 4, hipMemcpyDeviceToHost, s->stream));
    GOCTR_HIP(hipStreamSynchronize(s->stream));
    memcpy(y_out + o, s->h_out, (size_t)N * 4);
  }
  return 0;
}

}  // extern "C"
