// ctr_kernels.h -- device code of the DIN / YouTube-DNN step (float32, as in the reference):
// embedding gather + attention pooling, GEMM epilogues, attention backward, slab reduce, Adam.
//
// Reference arithmetic (file:line relative to auxten/go-ctr):
//   gather + row layout      recommend/rcmd.go:462-536, utils/util.go:22-28
//   cosine / euclid gate     model/activation.go:23-83, model/din/din.go:230-298
//   3 sigmoid layers+dropout model/din/din.go:301-315, model/youtube/dnn.go:162-177
//   BCE                      model/cost.go:9-17
//   Adam (gorgonia solver)   model/model.go:88,192  (semantics: SURVEY.md App. B)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ctr_x3_images.h"

namespace goctr {

// Per-step values that change between replays of the captured step graph live in device memory so
// that every kernel argument is a launch-time constant.
struct StepState {
  unsigned int gstep;     // global step counter: dropout hash input; Adam iter = gstep after ++
  unsigned int slot;      // next cost slot
  long long batch_idx;    // batch of the dataset the current step works on
  long long n_batches;
  // Adam's bias corrections of THIS step, 1 / (1 - beta^(gstep + 1)): two float64 pow calls that took 2.8 us of the reduce
  // blocks' critical path when each of them made them itself.  Written by whoever writes the state: the previous step's
  // loss block, or step_state_corr_kernel at the start of a call (same device pow, same bits).
  float corr1, corr2;
  float pcorr1, pcorr2;   // the corrections of the step BEFORE this state: what the split path's adam_kernel applies (it runs
                          // after the reduce has already advanced the state)
};
__device__ __forceinline__ void state_corrections(StepState& s, double beta1, double beta2) {
  const double it = (double)s.gstep + 1.0;
  s.corr1 = 1.0f / (float)(1.0 - pow(beta1, it));
  s.corr2 = 1.0f / (float)(1.0 - pow(beta2, it));
}

constexpr int COST_RING = 1 << 16;

// ---------------------------------------------------------------- scalar helpers
// gorgonia's float32 sigmoid clamps at -88 / +15 [from memory, SURVEY App. B]
// Hidden units and attention gates: v_exp_f32 / v_rcp_f32 based (|error| < 3e-7 absolute on
// [-88, 15]), two transcendental issues instead of the ~25-instruction IEEE expf + divide.
__device__ __forceinline__ float sigm_hidden(float x) {
  if (x < -88.f) return 0.f;
  if (x > 15.f) return 1.f;
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
// The output unit feeds log(1-p) in the BCE, which is ill-conditioned near saturation: evaluate it
// like the reference does (float64 exp, rounded once to float32).
__device__ __forceinline__ float sigm_out(float x) {
  if (x < -88.f) return 0.f;
  if (x > 15.f) return 1.f;
  return (float)(1.0 / (1.0 + exp((double)(-x))));
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// counter-hash dropout keep mask; bit-identical to oracle/orc_ctr.c:orc_dropout_keep
__device__ __forceinline__ float dropout_keep(uint32_t seed, uint32_t step, uint32_t layer, uint32_t row,
                                              uint32_t col, float p) {
  uint32_t h = mix32(seed ^ 0x9E3779B9u);
  h = mix32(h ^ (step * 2u + layer));
  h = mix32(h ^ row);
  h = mix32(h ^ (col * 0x85EBCA6Bu + 0xC2B2AE35u));
  float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  return u < (1.0f - p) ? 1.0f : 0.0f;
}

// where a batch's rows come from
struct RowSource {
  // dense-X mode (TrainSample layout, rcmd.go:56-63)
  const float* X; int xcols; int r_u, r_ub, r_v, r_c;
  // id mode (table + keys resident in HBM)
  const float* emb; long long V;
  const int32_t* ub_ids; const int32_t* item_ids; const float* ufeat; const float* cfeat;
  const float* Y;          // may be null (predict)
  long long rows;
  int id_mode;
  // key mode (serving passes, attn_fwd_kernel<..., KEYS = true>): the sample IS a (user, item, timestamp) key; behaviour ids,
  // side-feature rows and the candidate id are looked up where attn_fwd needs them -- recommend/rcmd.go:462-536
  // GetSampleVector + feature/ubcache/cache.go:71-94 TimeSeq.Filter inside the attention kernel, no assembled rows in HBM
  const int32_t* k_users; const int32_t* k_items; const long long* k_ts;
  const long long* ub_off; const int32_t* ub_items; const long long* ub_ts;     // behaviour cache CSR (ub_off null: none)
  const float* user_table; const float* item_table; long long n_users, n_items;
  unsigned char* k_failed;                                                        // [rows] 1 = key without features (scored as the zero row)
};

struct DropCfg {
  int mode; float p; const float* mask; int mask_ld; uint32_t seed; uint32_t layer; uint32_t row_off;
};

// ---------------------------------------------------------------- attention forward
struct AttnArgs {
  RowSource src;
  const StepState* st;
  int B, U, T, D, C, Ip;
  int kind, att;
  const float* att0;
  float* h0;    // [B, Ip]
  float* gate;  // [B, T]
  float* wgt;   // [B, T]
  // [B, T] or null.  Where the only reader of gate and weight is the chain launch's attention backward (DIN, frozen embeddings, id mode:
  // ctr.hip gate_fac_mode) the two leave as ONE factor (g (1 - g)) w -- what that backward multiplies its term with -- and gate / wgt
  // are not written: 2 x 1.6 MB less written by this launch and read by the chain launch at cfg3
  float* fac;
  int Tp_att;   // (reduce_attn_kernel) padded length of the att0 segment of the flat parameter buffer
  int xcd_affine;   // training launches: workgroup -> four-sample group by xcd_unit_of_block (below)
  float inv_T;      // 1.0f / (float)T, computed once on the host (the same IEEE division the kernels did per sample)
};

// Round 6: the attention's two square roots and its division as single v_sqrt_f32 / v_rcp_f32 issues (1 ulp each) instead of the
// correctly rounded ~9- and ~11-instruction sequences -- like sigm_hidden's exp / rcp since round 1.  A similarity weight moves by
// <= 3 ulp (4e-7), two orders below the 1e-5 the logits are held to; every attention variant shares attn_fwd_body, so all of this
// library's paths still agree bit for bit.
__device__ __forceinline__ float attn_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float attn_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// XCD affinity of batch rows (round 6, scripts/ubench/xcd_handoff.hip / profiles/r06_xcd_handoff.txt).  Workgroup b of a launch
// runs on XCD b % 8 (observed on every box so far; HIP does not promise it, so nothing here depends on it for correctness), and
// a tile that launch N wrote is read back by launch N + 1 in 0.4 of the time when the reader sits on the producer's XCD -- the
// lines are still in that XCD's L2 (18 KB: 1 030 against 2 200 cycles; 32 KB: 1 360 against 3 410); across XCDs they come
// from the Infinity Cache.  Every launch of the training step that produces or consumes BATCH ROWS therefore deals them by the
// same rule: rows are cut into granules of 128 (four 32-row tiles = one dW0 slab = 32 four-sample attention groups), granule g
// belongs to XCD g % 8.  unit = 32 (chain tiles per workgroup index) or 4-sample groups; `per` = units per granule.
// Only when the unit count is a multiple of 8 * per (else identity): B = 8192 / 16384 are, the odd test shapes are not.
__device__ __forceinline__ int xcd_unit_of_block(int b, int nunits, int per) {
  if (nunits % (8 * per) != 0) return b;
  const int x = b & 7, k = b >> 3;                 // the k-th workgroup of XCD x
  return per * (x + 8 * (k / per)) + k % per;     // granule x + 8 (k / per), unit k % per of it
}

// the same for a launch whose first `first` workgroups do something else (reduce_attn_kernel, adam_attn_kernel): v = the
// workgroup's index in the launch (XCD v % 8), the attention workgroups are v >= first
__device__ __forceinline__ int xcd_unit_of_block_after(int v, int first, int nunits, int per) {
  if (nunits % (8 * per) != 0) return v - first;
  const int x = v & 7;
  const int k = (v - (first + ((x - first) & 7))) >> 3;    // the k-th attention workgroup of XCD x
  return per * (x + 8 * (k / per)) + k % per;
}

// DPP lane exchanges (VALU, no LDS crossbar): quad_perm / row_half_mirror / row_mirror / row_ror
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the LPR consecutive lanes of a group (LPR = 1,2,4,8,16,32,64); every lane gets the sum
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  if (LPR >= 2) v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
  if (LPR >= 4) v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
  if (LPR >= 8) v += dpp_f32<0x141>(v);   // row_half_mirror: the other quad of the 8-lane half
  if (LPR >= 16) v += dpp_f32<0x140>(v);  // row_mirror: the other half of the 16-lane row
  if (LPR >= 32) v += __shfl_xor(v, 16, 64);
  if (LPR >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
// sum over the lanes that hold the same position (lane % LPR) in every group of the wavefront
template <int LPR>
__device__ __forceinline__ float cross_row_sum(float v) {
  if (LPR <= 1) v += dpp_f32<0xB1>(v);
  if (LPR <= 2) v += dpp_f32<0x4E>(v);
  if (LPR <= 4) v += dpp_f32<0x124>(v);   // row_ror:4  (position mod 4 preserved)
  if (LPR <= 8) v += dpp_f32<0x128>(v);   // row_ror:8
  if (LPR <= 16) v += __shfl_xor(v, 16, 64);
  if (LPR <= 32) v += __shfl_xor(v, 32, 64);
  return v;
}

// load VEC consecutive embedding lanes of one row (all-zero when the row pointer is null)
template <int VEC>
__device__ __forceinline__ void load_row(const float* row, int d0, int D, float x[VEC]) {
#pragma unroll
  for (int e = 0; e < VEC; ++e) x[e] = 0.f;
  if (row && d0 < D) {
    if (VEC == 4) {
      float4 t4 = *reinterpret_cast<const float4*>(row + d0);
      x[0] = t4.x; x[1] = t4.y; x[2] = t4.z; x[3] = t4.w;
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) x[e] = d0 + e < D ? row[d0 + e] : 0.f;
    }
  }
}

// same for a row pointer that is never null (id mode points missing ids at the all-zero row V of the device
// table, so the load needs no predicate and no zero-initialised destination); `full` = every lane's d0 < D
template <int VEC>
__device__ __forceinline__ void load_row_nn(const float* row, int d0, int D, bool full, float x[VEC]) {
  if (VEC == 4 && full) {
    float4 t4 = *reinterpret_cast<const float4*>(row + d0);
    x[0] = t4.x; x[1] = t4.y; x[2] = t4.z; x[3] = t4.w;
  } else {
    load_row<VEC>(row, d0, D, x);
  }
}

// a table row in the compile-time modes (FAST != 0): uniform base + a 32-bit byte offset per lane -- one multiply-add instead
// of a 64-bit multiply and add per gathered row (17 per sample).  The host only selects those modes for tables below 4 GB
// (ctr.hip attn_fast_mode); larger ones take the run-time-mode kernel with its 64-bit addresses.
template <int VEC>
__device__ __forceinline__ void load_row_off32(const float* base, int row, int D, int d0, float x[VEC]) {
  const unsigned off = ((unsigned)row * (unsigned)D + (unsigned)d0) * 4u;
  const float4 t4 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
  x[0] = t4.x; x[1 % VEC] = t4.y; x[2 % VEC] = t4.z; x[3 % VEC] = t4.w;
}

// One wavefront per sample.  A wavefront covers RPP = 64/LPR behaviour rows per pass, LPR lanes per
// row, VEC consecutive embedding lanes per lane (VEC=4 => one 16-byte load per lane, a 64-byte row
// of a D=16 table is fetched by 4 adjacent lanes: fully coalesced 64 B segments).  The ids of up to
// 64 behaviour slots are fetched with ONE coalesced load and handed to the row lanes by shuffle, then
// the row loads of NPB passes are issued back to back before any arithmetic, so the dependent chain
// is state -> ids -> rows (3 memory latencies) regardless of T.
// FAST: 0 = every mode decided at run time; 1 / 2 / 3 = id mode with D == LPR * VEC and the model kind fixed at
// compile time (YouTube mean pooling / DIN cosine / DIN euclid): the kernel is VALU-bound (rocprofv3: ~500 VALU
// instructions per sample against 10 loads and 6 stores), and run-time mode branches inside its unrolled loops cost
// more instructions than the arithmetic they select.
// (body: workgroup `blk` of the launch, the batch the step works on, the attention weights -- the merged
// reduce_attn_kernel below passes the NEXT step's batch index and weights it derived itself)
// what a workgroup of reduce_attn_kernel needs to pick up the attention weights its launch's reduce part is writing:
// the block that owns att0 publishes `expect` in *flag (device scope) once its stores are out (null: plain attn_fwd)
struct RaCtx { const unsigned int* flag; unsigned int expect; const float* att0; };

// sample: the launch's row this wavefront takes, or -1 = workgroup `blk`'s row of its wavefront index (4 per workgroup)
// key: (KEYS) the sample's key if its workgroup has already fetched it (ctr_serve16_kernel), else null = read it here
struct AttnKey { int user, item; long long ts; };
template <int VEC, int LPR, int FAST, bool KEYS = false>
__device__ __forceinline__ void attn_fwd_body(const AttnArgs& a, int blk, long long batch_idx, const float* att0w,
                                              const RaCtx* ra = nullptr, int sample = -1, const AttnKey* key = nullptr) {
  // (round 6: this body's sums of products may contract to fused multiply-adds -- one rounding instead of two per term, and 4 v_fma_f32
  // where the uncontracted code issued 2 v_pk_mul_f32 + 3 v_add_f32; the file's -ffp-contract=off stays for everything that must
  // match another path's bits.  Every attention-forward variant shares this body.)
#pragma clang fp contract(fast)
  const bool idm = FAST ? true : (bool)a.src.id_mode;
  const bool din = FAST ? FAST >= 2 : a.kind == GOCTR_DIN;
  const bool cosine = FAST ? FAST == 2 : a.att == GOCTR_ATT_COSINE;
  constexpr int RPP = 64 / LPR;
  constexpr int NPB = LPR < 4 ? LPR : 4;  // passes per block; NPB*RPP <= 64
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the sample's base addresses live in SGPRs
  const int b = sample >= 0 ? sample : blk * 4 + wave;
  if (b >= a.B) return;
  const RowSource& s = a.src;
  const long long gr = batch_idx * (long long)a.B + b;
  const bool valid = gr < s.rows;
  const int dl = lane % LPR, rl = lane / LPR;
  const int d0 = dl * VEC;
  const int D = a.D, T = a.T;

  // key mode: the sample's key -> its user / item feature rows and the window of its behaviour sequence
  const float* kurow = nullptr; const float* kirow = nullptr;     // feature rows (null: all-zero)
  int kitem = -1;                                                  // candidate id (-1: failed key => zero row)
  long long kb = 0, kcnt = 0;                                      // behaviour ids = ub_items[kb .. kb + kcnt)
  if (KEYS && valid) {
    // the key's three fields up front (they may sit in pinned HOST memory: one PCIe round trip, not three)
    const int u = key ? key->user : s.k_users[gr];
    const int it = key ? key->item : s.k_items[gr];
    const long long mts = key ? key->ts : (s.k_ts ? s.k_ts[gr] : 0);
    const bool ok = u >= 0 && u < s.n_users && it >= 0 && it < s.n_items;     // rcmd.go:291-307: else the ALL-zero row
    if (lane == 0 && s.k_failed) s.k_failed[gr] = ok ? 0 : 1;
    if (ok) {
      kurow = s.user_table + (long long)u * a.U; kirow = s.item_table + (long long)it * a.C; kitem = it;
      const long long b0 = s.ub_off ? s.ub_off[u] : 0, len = s.ub_off ? s.ub_off[u + 1] - b0 : 0;
      if (len > 0) {
        // TimeSeq.Filter (cache.go:71-94): first i with ts[i] <= maxTs in the newest-first sequence; maxTs == 0: from the newest
        long long lo = 0;
        if (mts != 0) {
          if (len <= 256) {
            lo = len;
            for (long long base = 0; base < len; base += 64) {
              const long long i = base + lane;
              const unsigned long long le = __ballot(i < len && s.ub_ts[b0 + i] <= mts);
              if (le) { lo = base + (long long)__builtin_ctzll(le); break; }
            }
          } else {
            long long hi = len;
            while (lo < hi) {
              const long long mid = (lo + hi) >> 1;
              if (s.ub_ts[b0 + mid] <= mts) hi = mid; else lo = mid + 1;
            }
          }
        }
        kb = b0 + lo;
        kcnt = len - lo < T ? len - lo : T;
      }
    }
  }
  // user / context side features: issue the loads first so they overlap the gather chain
  float uside[2] = {0.f, 0.f}, cside[2] = {0.f, 0.f};
  const bool side2 = a.U > 64 || a.C > 64;          // (wave-uniform: the reference's 52 / 53 columns fit one pass of the 64 lanes)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (k == 1 && !side2) break;
    const int ju = lane + 64 * k;
    if (KEYS) {
      uside[k] = (kurow && ju < a.U) ? kurow[ju] : 0.f;
      cside[k] = (kirow && ju < a.C) ? kirow[ju] : 0.f;
    } else {
      uside[k] = (valid && ju < a.U) ? (idm ? s.ufeat[gr * a.U + ju] : s.X[gr * (long long)s.xcols + s.r_u + ju]) : 0.f;
      cside[k] = (valid && ju < a.C) ? (idm ? s.cfeat[gr * a.C + ju] : s.X[gr * (long long)s.xcols + s.r_c + ju]) : 0.f;
    }
  }
  // the first 64 behaviour ids: requested BEFORE the candidate row is fetched and squared (both depend on nothing but the
  // sample's index; behind them the wave's chain was state -> item id -> item row -> ids -> rows, one latency longer)
  int ids64_first = -1;
  if (KEYS) { if (lane < kcnt) ids64_first = s.ub_items[kb + lane]; }
  else if (idm && valid && lane < T) ids64_first = s.ub_ids[gr * T + lane];
  // candidate item embedding v
  const bool full = FAST ? true : D == LPR * VEC;   // every lane owns VEC in-range embedding columns (wave-uniform)
  float vv[VEC];
  if (idm) {
    const int it = KEYS ? kitem : (valid ? s.item_ids[gr] : -1);
    if (FAST) load_row_off32<VEC>(s.emb, (it >= 0 && it < s.V) ? it : (int)s.V, D, d0, vv);
    else load_row_nn<VEC>(s.emb + (long long)((it >= 0 && it < s.V) ? it : s.V) * D, d0, D, full, vv);
  } else {
    load_row<VEC>(valid ? s.X + gr * (long long)s.xcols + s.r_v : nullptr, d0, D, vv);
  }
  float syy = 0.f;
#pragma unroll
  for (int e = 0; e < VEC; ++e) syy += vv[e] * vv[e];
  const float yn = attn_sqrt(group_sum<LPR>(syy));

  float psum[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) psum[e] = 0.f;

  // The ids of 64 slots arrive with one load; the rows are fetched NPB * RPP slots at a time, and with wide rows (D = 64:
  // 16 slots per block, 4 blocks per sample) the NEXT block's rows are requested before this block is pooled -- a sample
  // was a chain of four (ids, rows) round trips before, now it is ids + rows + three overlapped row fetches.
  constexpr int SLOTS = NPB * RPP;
  for (int tb0 = 0; tb0 < T; tb0 += 64) {
  int ids64 = ids64_first;
  if (tb0 > 0) {
    ids64 = -1;
    if (KEYS) { if (tb0 + lane < kcnt) ids64 = s.ub_items[kb + tb0 + lane]; }
    else if (idm && valid && tb0 + lane < T) ids64 = s.ub_ids[gr * T + tb0 + lane];
  }
  // (compile-time modes: a slot's id is checked ONCE, in its own lane -- missing, out of range and slots past T all become the
  // all-zero row V -- instead of once per pass behind the shuffle)
  if (FAST) ids64 = (unsigned)ids64 < (unsigned)s.V ? ids64 : (int)s.V;
  auto load_block = [&](int tbx, float (&xx)[NPB][VEC]) {
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
      const int t = tbx + p * RPP + rl;
      if (idm) {
        const int id = __shfl(ids64, (tbx - tb0) + p * RPP + rl, 64);
        if (FAST) load_row_off32<VEC>(s.emb, id, D, d0, xx[p]);
        else load_row_nn<VEC>(s.emb + (long long)((t < T && id >= 0 && id < s.V) ? id : s.V) * D, d0, D, full, xx[p]);
      } else {
        load_row<VEC>((valid && t < T) ? s.X + gr * (long long)s.xcols + s.r_ub + t * D : nullptr, d0, D, xx[p]);
      }
    }
  };
  constexpr int NB = 64 / SLOTS;
  float xb[NB][NPB][VEC];
#pragma unroll
  for (int k = 0; k < NB; ++k)
    if (k == 0 || tb0 + k * SLOTS < T) load_block(tb0 + k * SLOTS, xb[k]);
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const int tb = tb0 + k * SLOTS;
    if (tb >= T) break;
    float (&x)[NPB][VEC] = xb[k];
    // Row sums of every pass first; then slot tb + L's similarity / gate is computed ONCE, in lane L (the lanes of a
    // row group would otherwise all repeat the same sqrt / divide / exp once per pass), which also leaves gate and
    // weight in the layout of one coalesced store; the gate travels back to the row groups by shuffle.
    const int src = (lane % RPP) * LPR, pw = lane / RPP;
    float g_l = 1.0f, w_l = 0.f;
    if (din) {
      float s0 = 0.f, s1 = 0.f;      // cosine: (sxx, sxy) of slot tb + lane; euclid: (ss, -)
#pragma unroll
      for (int p = 0; p < NPB; ++p) {
        float u0 = 0.f, u1 = 0.f;
        if (cosine) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) { u0 += x[p][e] * x[p][e]; u1 += x[p][e] * vv[e]; }
          u0 = group_sum<LPR>(u0);
          u1 = group_sum<LPR>(u1);
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) { float df = x[p][e] - vv[e]; u0 += (d0 + e < D) ? df * df : 0.f; }
          u0 = group_sum<LPR>(u0);
        }
        const float v0 = __shfl(u0, src, 64), v1 = __shfl(u1, src, 64);
        if (pw == p) { s0 = v0; s1 = v1; }
      }
      if (cosine) {
        const float cosv = s1 * attn_rcp(attn_sqrt(s0) * yn + 1e-8f);
        w_l = (cosv + 1.0f) / 2.0f;
      } else {
        w_l = 1.0f - attn_sqrt(s0);
      }
      const int tl = tb + lane;
      float aw = 0.f;
      if (ra && ra->flag) {
        // merged kernel: the weights are being written by the att0 block of this very launch (dispatched before any
        // attention workgroup, so it is resident); wait for its flag, then read them past the non-coherent L2.  Relaxed
        // device-scope loads throughout: they go to memory themselves, and an acquire per poll would invalidate the L2
        // of 8192 polling wavefronts' XCDs over and over
        bool ok = true;
        if (tb == 0) {
          unsigned int spins = 0;
          while (__hip_atomic_load(ra->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ra->expect && ++spins < (1u << 22))
            __builtin_amdgcn_s_sleep(1);
          ok = spins < (1u << 22);           // (a wait that ran out poisons the step -- NaN costs -- instead of hanging the GPU)
          // compiler-level ordering only (no instruction, no L2 invalidate): the att0 loads below must not be hoisted
          // above the poll.  The hardware side is the sc1 loads themselves (they bypass L1 and are served past the
          // non-coherent L2) behind in-order vector memory.
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (lane < NPB * RPP && tl < T) aw = __hip_atomic_load(ra->att0 + tl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!ok) aw = __builtin_nanf("");
      } else if (lane < NPB * RPP && tl < T) {
        aw = att0w[tl];
      }
      g_l = sigm_hidden(w_l * aw);
    }
    if (a.fac) {
      if (lane < NPB * RPP && tb + lane < T) a.fac[(size_t)b * T + tb + lane] = (g_l * (1.0f - g_l)) * w_l;
    } else if (a.gate && lane < NPB * RPP && tb + lane < T) {      // (forward-only passes hand no gate / weight buffers: nobody reads them)
      a.gate[(size_t)b * T + tb + lane] = g_l;
      a.wgt[(size_t)b * T + tb + lane] = w_l;
    }
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
      const int t = tb + p * RPP + rl;
      const float g = din ? __shfl(g_l, p * RPP + rl, 64) : 1.0f;
      // (compile-time modes: a slot past T holds the all-zero row and a finite gate -- adding g * 0 leaves the same bits)
      if (FAST || t < T) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) psum[e] += g * x[p][e];
      }
    }
  }
  }
  // mean over T as a multiply by 1/T (one division per sample instead of one per column; <= 1 ulp from x / T)
  const float invT = a.inv_T;
  float* hrow = a.h0 + (size_t)b * a.Ip;
  if (VEC == 4 && (a.U & 3) == 0 && (a.Ip & 3) == 0 && (D & 3) == 0) {
    // pooled sum and candidate embedding as two 16-byte stores from the first row group
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 pv;
#pragma unroll
    for (int e = 0; e < 4; ++e) pv[e] = cross_row_sum<LPR>(psum[e]) * invT;
    if (rl == 0 && d0 < D) {
      *reinterpret_cast<v4*>(hrow + a.U + d0) = pv;
      *reinterpret_cast<v4*>(hrow + a.U + D + d0) = v4{vv[0], vv[1], vv[2], vv[3]};
    }
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float p = cross_row_sum<LPR>(psum[e]) * invT;
      if (rl == 0 && d0 + e < D) {
        hrow[a.U + d0 + e] = p;
        hrow[a.U + D + d0 + e] = vv[e];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (k == 1 && !side2) break;
    const int j = lane + 64 * k;
    if (j < a.U) hrow[j] = uside[k];
    if (j < a.C) hrow[a.U + 2 * D + j] = cside[k];
  }
  for (int j = lane + 128; j < a.U; j += 64)   // (side blocks wider than 128 columns)
    hrow[j] = KEYS ? (kurow ? kurow[j] : 0.f) : (valid ? (idm ? s.ufeat[gr * a.U + j] : s.X[gr * (long long)s.xcols + s.r_u + j]) : 0.f);
  for (int j = lane + 128; j < a.C; j += 64)
    hrow[a.U + 2 * D + j] = KEYS ? (kirow ? kirow[j] : 0.f) : (valid ? (idm ? s.cfeat[gr * a.C + j] : s.X[gr * (long long)s.xcols + s.r_c + j]) : 0.f);
}

template <int VEC, int LPR, int FAST>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  const int grp = a.xcd_affine ? xcd_unit_of_block((int)blockIdx.x, (a.B + 3) >> 2, 32) : (int)blockIdx.x;
  attn_fwd_body<VEC, LPR, FAST>(a, grp, a.st->batch_idx, a.att0);
}
// serving passes: the rows come from (user, item, timestamp) keys (RowSource key mode) -- key assembly and attention in one launch
template <int LPR, int FAST>
__global__ __launch_bounds__(256) void attn_fwd_keys_kernel(AttnArgs a) {
  attn_fwd_body<4, LPR, FAST, true>(a, (int)blockIdx.x, 0, a.att0);
}

// ---------------------------------------------------------------- attention backward (att0 grad)
struct AttnBwdArgs {
  RowSource src;
  const StepState* st;
  int B, T, D, Dp, Tp;
  const float* dp;    // [B, Dp]   d cost / d pooled
  const float* gate;  // [B, T]
  const float* wgt;   // [B, T]
  float* partial;     // dgs [B, Tp]: per-sample terms of the att0 gradient
};

constexpr int ATTN_BWD_WAVES = 16;  // samples per workgroup (one wavefront each)

// per-sample terms of  datt0[t] = sum_b dg[b,t] * g(1-g) * w ,  dg[b,t] = (1/T) sum_d dp[b,d] * x[b,t,d]
// (SURVEY A.1).  One wavefront per sample, all row gathers in flight at once; the terms go to
// dgs [B, Tp] and the sum over the batch is done by the weight-gradient GEMM launch as a ones-column
// product (same deterministic slab reduction as every other gradient).
template <int VEC, int LPR, int FAST>
__global__ __launch_bounds__(64 * ATTN_BWD_WAVES) void attn_bwd_kernel(AttnBwdArgs a) {
  const bool idm = FAST ? true : (bool)a.src.id_mode;
  constexpr int RPP = 64 / LPR;
  constexpr int NPB = LPR < 4 ? LPR : 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const RowSource& s = a.src;
  const int dl = lane % LPR, rl = lane / LPR, d0 = dl * VEC;
  const int D = a.D, T = a.T;
  const int b = blockIdx.x * ATTN_BWD_WAVES + wave;
  if (b >= a.B) return;
  const long long gr = a.st->batch_idx * (long long)a.B + b;
  const bool valid = gr < s.rows;
  float dpt[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) dpt[e] = d0 + e < D ? a.dp[(size_t)b * a.Dp + d0 + e] / (float)T : 0.f;
  float* out = a.partial + (size_t)b * a.Tp;
  for (int t = T + lane; t < a.Tp; t += 64) out[t] = 0.f;
  for (int tb = 0; tb < T; tb += NPB * RPP) {
    int myid = -1;
    if (idm && valid && lane < NPB * RPP && tb + lane < T) myid = s.ub_ids[gr * T + tb + lane];
    // gate / weight of slot tb + L arrive in lane L with one coalesced load each
    float gl = 0.f, wl = 0.f;
    if (lane < NPB * RPP && tb + lane < T) {
      gl = a.gate[(size_t)b * T + tb + lane];
      wl = a.wgt[(size_t)b * T + tb + lane];
    }
    float x[NPB][VEC];
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
      const int t = tb + p * RPP + rl;
      if (idm) {
        const int id = __shfl(myid, p * RPP + rl, 64);
        load_row_nn<VEC>(s.emb + (long long)((t < T && id >= 0 && id < s.V) ? id : s.V) * D, d0, D, FAST ? true : D == LPR * VEC, x[p]);
      } else {
        load_row<VEC>((valid && t < T) ? s.X + gr * (long long)s.xcols + s.r_ub + t * D : nullptr, d0, D, x[p]);
      }
    }
    const float gw = (gl * (1.0f - gl));     // g (1 - g) of slot tb + lane
    const int src = (lane % RPP) * LPR, pw = lane / RPP;
    float term = 0.f;
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
      float dg = 0.f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) dg += dpt[e] * x[p][e];
      dg = group_sum<LPR>(dg);
      const float dgs = __shfl(dg, src, 64);   // the row group of slot (p, lane % RPP)
      if (pw == p) term = dgs;
    }
    if (lane < NPB * RPP && tb + lane < T) out[tb + lane] = term * (gw * wl);   // one coalesced store (the factor first: AttnArgs::fac)
  }
}

// ---------------------------------------------------------------- GEMM epilogues
// hidden layer: P = sigmoid(z) ; A = P * mask / keep   (din.go:307-308,311-312)
struct EpiSigDrop {
  float* P; float* Aout; int ld; int ncols; DropCfg dr; const StepState* st;
  __device__ __forceinline__ void operator()(int row, int col, float z) const {
    float a = 0.f, post = 0.f;
    if (col < ncols) {
      a = sigm_hidden(z);
      post = a;
      if (dr.mode) {
        const float keep = 1.0f - dr.p;
        float m = dr.mode == 1 ? dr.mask[(size_t)row * dr.mask_ld + col]
                               : dropout_keep(dr.seed, st->gstep, dr.layer, dr.row_off + row, col, dr.p);
        post = a * m / keep;
      }
    }
    P[(size_t)row * ld + col] = a;
    if (Aout != P) Aout[(size_t)row * ld + col] = post;
  }
};

// output unit + BCE + d cost/d z2 (din.go:315, cost.go:9-17).  Works on the padded 16-column tile;
// only column 0 is real.
struct EpiOut {
  float* yhat;      // [B]
  float* lossrow;   // [B] or null (predict)
  float* dz2;       // [B,16] or null
  const float* Y; long long rows; const StepState* st; int B; float inv_bglobal;
  __device__ __forceinline__ void operator()(int row, int col, float z) const {
    if (col != 0) {
      if (dz2) dz2[(size_t)row * 16 + col] = 0.f;
      return;
    }
    const float p = sigm_out(z);
    yhat[row] = p;
    if (!lossrow) return;
    const long long gr = st->batch_idx * (long long)B + row;
    const float y = (Y && gr < rows) ? Y[gr] : 0.f;  // pad rows: y = 0 (model.go:180-184)
    const float one_eps = (float)(1.0 + 1e-8);      // == 1.0f (quirk Q2)
    const float positive = logf(p) * y;
    const float negative = logf(one_eps - p) * (1.0f - y);
    lossrow[row] = positive + negative;
    const float dy = -((y / p) - ((1.0f - y) / (one_eps - p))) * inv_bglobal;
    dz2[(size_t)row * 16] = dy * (p * (1.0f - p));
  }
};

// backward data: dz = (delta . W^T) * (mask/keep) * P(1-P)
struct EpiDSig {
  float* out; const float* P; int ld; int ncols; DropCfg dr; const StepState* st;
  __device__ __forceinline__ void operator()(int row, int col, float s) const {
    float r = 0.f;
    if (col < ncols) {
      float k = 1.0f;
      if (dr.mode) {
        const float keep = 1.0f - dr.p;
        float m = dr.mode == 1 ? dr.mask[(size_t)row * dr.mask_ld + col]
                               : dropout_keep(dr.seed, st->gstep, dr.layer, dr.row_off + row, col, dr.p);
        k = m / keep;
      }
      const float a = P[(size_t)row * ld + col];
      r = (s * k) * (a * (1.0f - a));
    }
    out[(size_t)row * ld + col] = r;
  }
};

struct EpiStore {
  float* out; int ld;
  __device__ __forceinline__ void operator()(int row, int col, float v) const { out[(size_t)row * ld + col] = v; }
};

// ---------------------------------------------------------------- slab reduce
struct ReduceSeg { const float* slabs; int nslabs; unsigned long long stride; int begin; int len; };
struct ReduceArgs {
  ReduceSeg seg[4];
  int nseg; int nflat;
  const float* lossrow; int B;
  float* G;           // [nflat + 1]; G[nflat] = sum of per-row BCE terms
  const StepState* st;   // the step's state (read by every kernel of the step)
  StepState* st_out;     // advance == 1: where the NEXT step's state is written (the other ping-pong slot, so
                         // no kernel of this step can observe the update: no inter-workgroup ordering needed)
  int advance;        // 1: this launch closes the step (++gstep, next batch)
};

__device__ __forceinline__ void advance_state(const StepState* in, StepState* out) {
  StepState s = *in;
  s.gstep += 1;
  s.slot += 1;
  const long long nb = s.batch_idx + 1;
  s.batch_idx = nb >= s.n_batches ? 0 : nb;
  s.pcorr1 = in->corr1; s.pcorr2 = in->corr2;     // (corr1 / corr2 of the new state: adam_kernel's extra block)
  *out = s;
}

// deterministic sum of the per-row BCE terms by ONE workgroup: all loads of a thread are issued before the
// (fixed-order) adds -- one memory latency, not B/256 -- then a fixed-shape tree in LDS; result in red[0]
__device__ __forceinline__ void loss_sum_block(const float* lossrow, int B, float* red) {
  float s = 0.f;
  for (int i0 = threadIdx.x * 4; i0 < B; i0 += 256 * 4 * 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 1024;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i + 3 < B) v[u] = *reinterpret_cast<const float4*>(lossrow + i);
      else if (i < B) { v[u].x = lossrow[i]; if (i + 1 < B) v[u].y = lossrow[i + 1]; if (i + 2 < B) v[u].z = lossrow[i + 2]; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
}

// Every group of 4 consecutive gradient entries is summed by 8 adjacent lanes (lane p takes a
// contiguous 1/8 of the slabs, 16-byte loads, all independent => deep memory-level parallelism),
// partial sums are combined by xor-shuffles: a fixed association order => bitwise reproducible.
#ifndef GOCTR_NO_PLAIN_KERNELS   // (a second translation unit includes this header for its templates only: ctr_fwd.hip)
__global__ __launch_bounds__(256) void reduce_kernel(ReduceArgs a) {
  if (blockIdx.x == gridDim.x - 1) {  // last block: deterministic loss sum
    __shared__ float red[256];
    loss_sum_block(a.lossrow, a.B, red);
    if (threadIdx.x == 0) {
      a.G[a.nflat] = red[0];
      if (a.advance) advance_state(a.st, a.st_out);
    }
    return;
  }
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int e0 = (gid >> 3) * 4, p = gid & 7;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e0 < a.nflat) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < a.nseg && e0 >= a.seg[k].begin && e0 < a.seg[k].begin + a.seg[k].len) {
        const int spp = (a.seg[k].nslabs + 7) >> 3;
        int lo = p * spp, hi = lo + spp;
        if (hi > a.seg[k].nslabs) hi = a.seg[k].nslabs;
        const float* base = a.seg[k].slabs + (e0 - a.seg[k].begin);
#pragma unroll 8
        for (int j = lo; j < hi; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(base + (size_t)j * a.seg[k].stride);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      }
    }
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
    acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
  }
  if (p == 0 && e0 < a.nflat) *reinterpret_cast<float4*>(a.G + e0) = acc;
}
#endif

// ---------------------------------------------------------------- Adam (gorgonia AdamSolver.Step)
struct AdamArgs {
  float* W; float* G; float* Mo; float* Vo; int nflat;
  // layout of the flat parameter buffer
  int off1, off2, offa;      // W0 at 0, W1 at off1, W2 at off2, att0 at offa
  int Ip, H1p, H2p, Dp, U, D;
  float* W1T; float* W2T; float* W0sT;  // transposed copies used by the backward-data GEMMs
  float* W0i; float* W1i; float* W1Ti; float* W0sTi;  // LDS images [K/4][N][4] read by the chain kernel
  float* W0pvT; int Npv;     // trainable embeddings (emb_train.h): W0[U : U+2D, :]^T as [H1p, Npv], the B operand of dpv = dz0 . W0pv^T (null: not kept)
  CxImages x3;               // bf16-plane fragment images of the 6-product-split chain kernel (img0 == null: not kept)
  double lr, l2, beta1, beta2, eps;
  int div_by_batch, l2_first;
  int bglobal;
  const StepState* st;
  float* costs;              // ring [COST_RING]
};

// index of element (k, n) of a [K x N] operand inside its LDS image [K/4][N][4] (ctr_chain.h)
__host__ __device__ inline size_t img_index(int k, int n, int N) { return ((size_t)(k >> 2) * N + n) * 4 + (k & 3); }

// one parameter's gorgonia-order Adam update (+ transposed operand copies)
// w0 / m0 / v0: the element's weight and moments, loaded by the caller (the fused kernel requests them before it walks
// the gradient slabs, so that they do not cost a second memory round trip after the reduction)
__device__ __forceinline__ void adam_apply_pre(const AdamArgs& a, int idx, float g, float corr1, float corr2, float w0, float m0, float v0);
__device__ __forceinline__ void adam_apply(const AdamArgs& a, int idx, float g, float corr1, float corr2) {
  adam_apply_pre(a, idx, g, corr1, corr2, a.W[idx], a.Mo[idx], a.Vo[idx]);
}
// the arithmetic of one element's update, without side effects (reduce_attn_kernel recomputes the attention weights of
// the next step with it: same expressions, same bits)
__device__ __forceinline__ float adam_new_weight(const AdamArgs& a, float g, float corr1, float corr2, float w0, float m0, float v0,
                                                 float& m, float& v) {
  const float b1 = (float)a.beta1, b2 = (float)a.beta2;
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  const float l2 = (float)a.l2, eps = (float)a.eps, neg_eta = (float)(-a.lr);
  const float one_per_batch = 1.0f / (float)a.bglobal;
  if (a.l2_first) {
    if (l2 != 0.f) g = g + w0 * l2;
    if (a.div_by_batch && a.bglobal > 1) g = g * one_per_batch;
  } else {
    if (a.div_by_batch && a.bglobal > 1) g = g * one_per_batch;
    if (l2 != 0.f) g = g + w0 * l2;
  }
  const float t1 = omb1 * g;
  const float g2 = (g * g) * omb2;
  m = b1 * m0 + t1;
  v = b2 * v0 + g2;
  const float mhat = m * corr1;
  const float vhat = v * corr2;
  return w0 + (neg_eta * mhat) / (sqrtf(vhat) + eps);
}
__device__ __forceinline__ void adam_apply_pre(const AdamArgs& a, int idx, float g, float corr1, float corr2, float w0, float m0, float v0) {
  float w = w0;
  float m, v;
  w = adam_new_weight(a, g, corr1, corr2, w0, m0, v0, m, v);
  a.Mo[idx] = m; a.Vo[idx] = v;
  a.W[idx] = w;
  // keep the transposed operand copies in sync
  if (idx < a.off1) {
    const int r = idx / a.H1p, c = idx - r * a.H1p;
    a.W0i[img_index(r, c, a.H1p)] = w;
    if (r >= a.U && r < a.U + a.D) {
      a.W0sT[(size_t)c * a.Dp + (r - a.U)] = w;
      a.W0sTi[img_index(c, r - a.U, a.Dp)] = w;
    }
    if (a.W0pvT && r >= a.U && r < a.U + 2 * a.D) a.W0pvT[(size_t)c * a.Npv + (r - a.U)] = w;
  } else if (idx < a.off2) {
    const int k = idx - a.off1;
    const int r = k / a.H2p, c = k - r * a.H2p;
    a.W1T[(size_t)c * a.H1p + r] = w;
    a.W1i[img_index(r, c, a.H2p)] = w;
    a.W1Ti[img_index(c, r, a.H1p)] = w;
  } else if (idx < a.offa) {
    const int k = idx - a.off2;
    const int r = k / 16, c = k - r * 16;
    a.W2T[(size_t)c * a.H2p + r] = w;
  }
  cx_scatter_weight(a.x3, w, idx, a.off1, a.off2, a.H1p, a.H2p, a.U, a.D);
}

// (split path: the reduce kernel has already advanced the state.  This step's bias corrections are the new state's pcorr;
// one extra block computes the new state's own -- the next step's -- off everybody's critical path)
#ifndef GOCTR_NO_PLAIN_KERNELS   // (a second translation unit includes this header for its templates only: ctr_fwd.hip)
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  if (blockIdx.x == gridDim.x - 1) {
    if (threadIdx.x == 0) {
      StepState* w = const_cast<StepState*>(a.st);
      StepState t = *w;
      state_corrections(t, a.beta1, a.beta2);
      w->corr1 = t.corr1; w->corr2 = t.corr2;
    }
    return;
  }
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const float corr[2] = {a.st->pcorr1, a.st->pcorr2};
  if (idx == 0) {
    const float s = a.G[a.nflat];
    a.costs[(a.st->slot - 1u) % COST_RING] = -(s / (float)a.bglobal);  // cost.go:15 Neg(Mean(...))
  }
  if (idx >= a.nflat) return;
  const float g = a.G[idx];
  a.G[idx] = 0.f;
  adam_apply(a, idx, g, corr[0], corr[1]);
}
#endif

// Single-GPU fast path: slab reduce + Adam + BCE sum + step advance in ONE launch.
struct ReduceAdamArgs {
  ReduceArgs r; AdamArgs ad;
  unsigned int* ra_flag; int ra_block;    // reduce_attn_kernel: block ra_block owns the att0 segment and publishes gstep + 1 (-1: none)
  int skip_begin, skip_len;               // parameters [skip_begin, skip_begin + skip_len) were already updated by an earlier launch of
                                          // this step (ctr_chain_x3.h att0_early_body: att0 inside the weight-gradient launch): hands off
};

// sum of one float4 parameter group (elements e0 .. e0 + 3) over the slabs of its segment: the 8 lanes of the group take a
// contiguous eighth of the slabs each, a xor-butterfly leaves the total in all of them.  Every lane of the 8 must call it.
// Up to 8 slabs per lane are loaded before the first add (one memory latency); the adds keep the order of the plain loop.
__device__ __forceinline__ float4 slab_group_sum(const ReduceArgs& a, int e0, int pl) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* base = a.seg[0].slabs;
  unsigned long long stride = 0;
  int lo = 0, hi = 0;
  if (e0 < a.nflat) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < a.nseg && e0 >= a.seg[k].begin && e0 < a.seg[k].begin + a.seg[k].len) {
        const int spp = (a.seg[k].nslabs + 7) >> 3;
        lo = pl * spp; hi = lo + spp;
        if (hi > a.seg[k].nslabs) hi = a.seg[k].nslabs;
        base = a.seg[k].slabs + (e0 - a.seg[k].begin);
        stride = a.seg[k].stride;
      }
    }
  }
  if (__all(hi - lo <= 8)) {                                  // (wave-uniform choice of the path)
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = lo + u < hi ? lo + u : (hi > lo ? hi - 1 : 0);
      v[u] = (hi > lo) ? *reinterpret_cast<const float4*>(base + (size_t)j * stride) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (lo + u < hi) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  } else {
#pragma unroll 8
    for (int j = lo; j < hi; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(base + (size_t)j * stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
    acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
  }
  return acc;
}

__device__ __forceinline__ void reduce_adam_body(const ReduceAdamArgs& p, int blk_in, int nblk) {
  const ReduceArgs& a = p.r;
  // merged launch: the block that owns att0 trades places with block 0 -- dispatched first, its loads are not queued behind
  // the 10 MB of slab reads of the other 376 blocks
  const int blk = p.ra_block < 0 ? blk_in : (blk_in == 0 ? p.ra_block : (blk_in == p.ra_block ? 0 : blk_in));
  __shared__ float red[256];
  if (blk == nblk - 1) {   // the loss block runs beside the gradient blocks
    // the next step's state, bias corrections included (thread 0's pow calls run under the loss rows' load latency)
    StepState ns = *a.st;
    if (threadIdx.x == 0) {
      ns.gstep += 1; ns.slot += 1;
      const long long nb = ns.batch_idx + 1;
      ns.batch_idx = nb >= ns.n_batches ? 0 : nb;
      ns.pcorr1 = ns.corr1; ns.pcorr2 = ns.corr2;
      state_corrections(ns, p.ad.beta1, p.ad.beta2);
    }
    loss_sum_block(a.lossrow, a.B, red);
    if (threadIdx.x == 0) {
      a.G[a.nflat] = red[0];
      p.ad.costs[a.st->slot % COST_RING] = -(red[0] / (float)p.ad.bglobal);   // cost.go:15 Neg(Mean(...))
      *a.st_out = ns;
    }
  } else {
    const int gid = blk * 256 + threadIdx.x;
    const int e0 = (gid >> 3) * 4, pl = gid & 7;
    const bool mine = pl < 4 && e0 + pl < a.nflat &&         // lanes 0..3 of a group own one parameter each
                      !(e0 + pl >= p.skip_begin && e0 + pl < p.skip_begin + p.skip_len);
    float w0 = 0.f, m0 = 0.f, v0 = 0.f;
    if (mine) { w0 = p.ad.W[e0 + pl]; m0 = p.ad.Mo[e0 + pl]; v0 = p.ad.Vo[e0 + pl]; }
    const float corr1 = a.st->corr1, corr2 = a.st->corr2;    // (the step's bias corrections travel with its state)
    const float4 acc = slab_group_sum(a, e0, pl);
    // after the butterfly all 8 lanes hold the sums: lanes 0..3 of the group update one element each
    if (mine) {
      const float g = pl == 0 ? acc.x : (pl == 1 ? acc.y : (pl == 2 ? acc.z : acc.w));
      adam_apply_pre(p.ad, e0 + pl, g, corr1, corr2, w0, m0, v0);
    }
    if (blk == p.ra_block) {
      // the attention workgroups of this launch wait for the new att0.  A release fence here would write back every dirty
      // line of this XCD's L2 -- while 2000 workgroups are storing h0 through it: each thread instead repeats its own
      // element's store as a device-scope (write-through) one, the barrier's s_waitcnt sees those acknowledged, and only
      // then does thread 0 publish the flag, also write-through (measured: 52.3 us per step against 52.8 with __threadfence).
      // The readers use device-scope loads.
      if (mine) {
        const float wn = p.ad.W[e0 + pl];
        __hip_atomic_store(p.ad.W + e0 + pl, wn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (compiler-level: the flag store must not be moved above the barrier)
      if (threadIdx.x == 0) __hip_atomic_store(p.ra_flag, a.st->gstep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// the bias corrections of the state a call starts from (set_state / a restored checkpoint / a retargeted cursor leave them
// to this launch; inside a call every step's loss block writes the next state's)
#ifndef GOCTR_NO_PLAIN_KERNELS   // (a second translation unit includes this header for its templates only: ctr_fwd.hip)
__global__ void step_state_corr_kernel(StepState* st, double beta1, double beta2) {
  StepState s = *st;
  state_corrections(s, beta1, beta2);
  st->corr1 = s.corr1; st->corr2 = s.corr2;
}
#endif

#ifndef GOCTR_NO_PLAIN_KERNELS   // (a second translation unit includes this header for its templates only: ctr_fwd.hip)
__global__ __launch_bounds__(256) void reduce_adam_kernel(ReduceAdamArgs p) {
  reduce_adam_body(p, (int)blockIdx.x, (int)gridDim.x);
}
#endif

// ---------------------------------------------------------------- the step's last launch merged with the NEXT step's first
// reduce_adam (blocks [0, nred)) beside attn_fwd of the following batch (the other blocks).  The two do not depend on each
// other -- except through the step state (the attention part derives the next batch cursor itself, like advance_state) and
// the attention weights att0 (T floats, DIN): the reduce block that owns them publishes a flag after its Adam update and an
// attention wavefront waits for it right before its gate, ~2 us into its life (the reduce blocks have the lowest block
// indices: they are resident before any attention workgroup is dispatched; the wait is bounded all the same).  The eight
// L2s are not coherent: release fence + device-scope flag on the writer's side, device-scope loads of the 50 floats on the
// reader's.  reduce_adam is a latency chain on a few hundred small workgroups and attn_fwd one on 2048: side by side they
// cost little more than the longer one, and the step loses a launch boundary.
// tests/test_gpu_ctr.py::test_graph_replay_equals_eager compares this (graph) path with the unmerged eager one bit by bit.
template <int VEC, int LPR, int FAST>
__global__ __launch_bounds__(256) void reduce_attn_kernel(ReduceAdamArgs p, AttnArgs a, int nred) {
  if ((int)blockIdx.x < nred) { reduce_adam_body(p, (int)blockIdx.x, nred); return; }
  const ReduceArgs& r = p.r;
  long long nb = r.st->batch_idx + 1;                          // advance_state()'s cursor
  if (nb >= r.st->n_batches) nb = 0;
  const RaCtx ctx{p.ra_flag, r.st->gstep + 1u, a.att0};
  const int grp = a.xcd_affine ? xcd_unit_of_block_after((int)blockIdx.x, nred, (a.B + 3) >> 2, 32) : (int)blockIdx.x - nred;
  attn_fwd_body<VEC, LPR, FAST>(a, grp, nb, a.att0, FAST >= 2 ? &ctx : nullptr);      // (ctx.flag == null: att0 is already this step's, no wait)
}

// The data-parallel step's counterpart of reduce_attn_kernel: the step is [.. reduce] -> all-reduce -> [Adam], and what can share
// a launch with the next step's attention is the part BEHIND the collective.  Blocks [0, nadam) are adam_kernel's (the reduce
// has already advanced the state: `ad.st` is the NEW state, whose cursor is the batch the attention part gathers), the att0
// block trades places with block 0 and publishes the new state's gstep once its weights are out, exactly like the reduce
// block of reduce_attn_kernel.  Same arithmetic as adam_kernel + attn_fwd_kernel: tests/test_gpu_comm.py holds the
// data-parallel path (one-rank communicator) to the single-GPU path's bits.
template <int VEC, int LPR, int FAST>
__global__ __launch_bounds__(256) void adam_attn_kernel(AdamArgs ad, unsigned int* ra_flag, int ra_block, AttnArgs a, int nadam) {
  const int bi = (int)blockIdx.x;
  if (bi < nadam) {
    const int blk = ra_block < 0 ? bi : (bi == 0 ? ra_block : (bi == ra_block ? 0 : bi));
    if (blk == nadam - 1) {                      // (adam_kernel's extra block: the new state's own bias corrections)
      if (threadIdx.x == 0) {
        StepState* w = const_cast<StepState*>(ad.st);
        StepState t = *w;
        state_corrections(t, ad.beta1, ad.beta2);
        w->corr1 = t.corr1; w->corr2 = t.corr2;
      }
      return;
    }
    const int idx = blk * 256 + threadIdx.x;
    const float c1 = ad.st->pcorr1, c2 = ad.st->pcorr2;
    if (idx == 0) {
      const float sum = ad.G[ad.nflat];
      ad.costs[(ad.st->slot - 1u) % COST_RING] = -(sum / (float)ad.bglobal);  // cost.go:15 Neg(Mean(...))
    }
    if (idx < ad.nflat) {
      const float g = ad.G[idx];
      ad.G[idx] = 0.f;
      adam_apply(ad, idx, g, c1, c2);
    }
    if (blk == ra_block) {                       // (see reduce_adam_body: write-through repeats of the own stores, then the flag)
      if (idx < ad.nflat) {
        const float wn = ad.W[idx];
        __hip_atomic_store(ad.W + idx, wn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (threadIdx.x == 0) __hip_atomic_store(ra_flag, ad.st->gstep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const RaCtx ctx{ra_flag, ad.st->gstep, a.att0};
  const int grp = a.xcd_affine ? xcd_unit_of_block_after(bi, nadam, (a.B + 3) >> 2, 32) : bi - nadam;
  attn_fwd_body<VEC, LPR, FAST>(a, grp, ad.st->batch_idx, a.att0, FAST >= 2 ? &ctx : nullptr);
}

// ---------------------------------------------------------------- standalone gather (bit-exact)
struct GatherArgs {
  const float* emb; long long V; int D, T, U, C;
  const int32_t* ub_ids; const int32_t* item_ids; const float* ufeat; const float* cfeat;
  long long rows; float* X; int xcols;
};
// rcmd.go:497-533: one wavefront per row; pure copies => bit-exact
#ifndef GOCTR_NO_PLAIN_KERNELS   // (a second translation unit includes this header for its templates only: ctr_fwd.hip)
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r = (long long)blockIdx.x * 4 + wave;
  if (r >= a.rows) return;
  float* row = a.X + r * a.xcols;
  for (int j = lane; j < a.U; j += 64) row[j] = a.ufeat[r * a.U + j];
  const int TD = a.T * a.D;
  for (int j = lane; j < TD + a.D; j += 64) {
    const int t = j / a.D, d = j - t * a.D;
    const int id = t < a.T ? a.ub_ids[r * a.T + t] : a.item_ids[r];
    row[a.U + j] = (id >= 0 && id < a.V) ? a.emb[(long long)id * a.D + d] : 0.f;
  }
  for (int j = lane; j < a.C; j += 64) row[a.U + TD + a.D + j] = a.cfeat[r * a.C + j];
}
#endif

}  // namespace goctr
