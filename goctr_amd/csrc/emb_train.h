// Trainable item embeddings: sparse gradient + SGD scatter-add (EXTENSION — SURVEY F3 / K18 / 8(e) "Trainable
// embeddings (extension, cfg4)", BASELINE north_star "embedding gather and SGD scatter-add").
//
// The reference keeps the embedding table frozen while DIN / YouTube-DNN train (din.go:161-169 and dnn.go:152-154
// list the learnables), so there is no reference code behind this file; its oracle is oracle/orc_embtrain.c (float64,
// finite-difference checked).  Math (per live sample b, slot t with a valid id, T slots, dp / dvh = d cost / d pooled
// and d cost / d item segment of h0):
//     dx_t = (g_t / T) dp + q_t dw_t/dx_t          q_t = ((dp . x_t) / T) g_t (1 - g_t) att0[t]        (DIN; YouTube: g = 1, q = 0)
//     dv   = dvh + sum_t q_t dw_t/dv
//     cosine: dw/dx = (v/den - s |v| x / (|x| den^2)) / 2,  dw/dv = (x/den - s |x| v / (|v| den^2)) / 2,  den = |x||v| + 1e-8
//     euclid: dw/dx = -(x - v)/|x - v|,  dw/dv = (x - v)/|x - v|
//     E[id] -= lr * (sum of the row gradients of every occurrence of id in the batch)
//
// Launch sequence per step (all on the step's stream, inside the step's hipGraph):
//   emb_mark(2)   which ids does the batch touch, and (single GPU, vocabulary larger than the batch's id count) which of
//                 them exactly once: those rows are updated in place by emb_grad, no accumulator, no emb_apply work
//   scan x3       rank = exclusive prefix sum over the vocabulary: the ids that accumulate get dense slots 0..n-1 in
//                 ASCENDING ID ORDER (deterministic numbering, no hashing, no sort); the last pass also writes the
//                 slot -> id list and clears the marks
//   w0pv_t, gemm  dpv[B, 2D] = dz0 . W0[U : U+2D, :]^T   (MFMA; dz0 is what the chain kernel already stored)
//   emb_grad      one lane group per (sample, slot): row gradient -> 64-bit FIXED-POINT atomic adds, hot rows staged in an
//                 LDS cache per workgroup, the rest straight into accum[rank[id]]
//   emb_apply     E[id] -= lr * accum ; clears accum behind itself
// Fixed point (2^-44 units, range +-5e5) makes the scatter-add associative: the updated table is bit-identical from
// run to run whatever order the atomics land in — the same reproducibility contract as the dense weights' slab
// reduction — at a resolution (5.7e-14) far below fp32's for gradients of this size.
#pragma once
#include <hip/hip_runtime.h>

#include "ctr_kernels.h"

namespace goctr {

constexpr double EMB_FIX_SCALE = 17592186044416.0;          // 2^44
constexpr double EMB_FIX_INV = 1.0 / 17592186044416.0;

struct EmbTrainArgs {
  RowSource src;
  const StepState* st;
  int B, T, D, kind, att;
  const float* dpv; int ldp;       // [B, ldp]: columns 0..D-1 = d cost / d pooled, D..2D-1 = d cost / d item embedding
  const float* gate; const float* wgt; const float* att0;
  float* emb; long long V;
  unsigned int* mark;              // [W * Vw], indexed by emb_pidx(id)
  const unsigned int* rank;        // [W * Vw], indexed by emb_pidx(id)
  // Data parallel (W ranks, replicated table): the mark / rank arrays are indexed OWNER-MAJOR, pidx(id) = (id % W) * Vw +
  // id / W with Vw = round_up(ceil(V / W), 4), so that the rank scan numbers the touched ids bucket after bucket
  // (owner = id % W) in ascending id order: a rank's gradient rows for owner o are one contiguous range of accum --
  // the send buffer of the bucketed exchange (ctr.hip: launch_emb_exchange), no packing pass.  W == 1: pidx(id) == id.
  int W; long long Vw;
  long long* accum;                // [min(V, B (T+1)), D]
  float lr;
  int dbg;                         // timing experiments only (always 0 in the library; scripts/ubench builds set bits): 1 skip the flush, 2 skip cache misses, 4 skip LDS adds
};

constexpr unsigned int EMB_MULTI = 0xFFFFFFFFu;

__device__ __forceinline__ long long emb_pidx(const EmbTrainArgs& a, int id) {
  if (a.W == 1) return id;
  const int q = id / a.W;
  return (long long)(id - q * a.W) * a.Vw + q;
}

// mark[id]: 0 = untouched, EMB_MULTI = accumulate (gets a slot), p + 1 = touched by pair p only ("single").
// singles == 0: every touched id is EMB_MULTI.  singles == 1: first pass — the last writer's pair index stays;
// second pass (emb_mark2) — a pair that does not find its own index there knows the id has at least two pairs and
// overwrites it with EMB_MULTI.  Which pair wins the first pass is arbitrary, the outcome is not: EMB_MULTI iff >= 2 pairs.
__device__ __forceinline__ int emb_pair_id(const EmbTrainArgs& a, long long p) {
  const int per = a.T + 1;
  if (p >= (long long)a.B * per) return -1;
  const int b = (int)(p / per), t = (int)(p % per);
  const long long gr = a.st->batch_idx * (long long)a.B + b;
  if (gr >= a.src.rows) return -1;                                // padded row: no ids
  const int id = t < a.T ? a.src.ub_ids[gr * a.T + t] : a.src.item_ids[gr];
  return (id >= 0 && id < a.V) ? id : -1;
}

__global__ void emb_mark_kernel(EmbTrainArgs a, int singles) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const int id = emb_pair_id(a, p);
  if (id >= 0) a.mark[emb_pidx(a, id)] = singles ? (unsigned int)p + 1u : EMB_MULTI;
}

__global__ void emb_mark2_kernel(EmbTrainArgs a) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const int id = emb_pair_id(a, p);
  if (id >= 0 && a.mark[emb_pidx(a, id)] != (unsigned int)p + 1u) a.mark[emb_pidx(a, id)] = EMB_MULTI;
}

// W0pvT[k][n] = W0[U + n][k], n < 2D (zero beyond): the B operand of dpv = dz0 . W0[U:U+2D, :]^T
__global__ void w0pv_transpose_kernel(const float* W0, int H1p, int U, int D2, int Np, float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H1p * Np) return;
  const int k = i / Np, n = i % Np;
  out[i] = n < D2 ? W0[(size_t)(U + n) * H1p + k] : 0.f;
}

template <int GS>
__device__ __forceinline__ float emb_group_sum(float v) {
#pragma unroll
  for (int o = 1; o < GS; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// integer twin of group_sum (ctr_kernels.h): DPP inside a 16-lane row, LDS crossbar only across rows
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int GS>
__device__ __forceinline__ int emb_group_sum_int(int v) {
  if (GS >= 2) v += dpp_i32<0xB1>(v);
  if (GS >= 4) v += dpp_i32<0x4E>(v);
  if (GS >= 8) v += dpp_i32<0x141>(v);
  if (GS >= 16) v += dpp_i32<0x140>(v);
  if (GS >= 32) v += __shfl_xor(v, 16, 64);
  if (GS >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// rint(g * 2^44) without float64 / 64-bit conversions (both are multi-instruction sequences here): g 2^20 = hi + rem with hi
// = rint(g 2^20) and rem exact in float32, so g 2^44 = hi 2^24 + rem 2^24 exactly and rint() only touches the second term.
// |g| is clamped to 2^10 (a row gradient of that size is a diverged run anyway).
__device__ __forceinline__ long long emb_fix(float g) {
  g = g == g ? fminf(fmaxf(g, -1024.0f), 1024.0f) : 0.0f;       // (a NaN gradient updates nothing)
  const float s = g * 1048576.0f;
  const float hi = rintf(s);
  const float lo = rintf((s - hi) * 16777216.0f);
  return ((long long)(int)hi << 24) + (long long)(int)lo;
}

// LDS staging of hot rows: item popularity is Zipfian, and global atomics on one row serialise (the same effect as in
// the dictionary build, corpus.hip).  Each workgroup owns a direct-mapped cache of NSLOT rows of 64-bit accumulators in
// LDS: the first id that hashes to a slot claims it; its later occurrences inside this workgroup are LDS atomics, every
// other id of that slot goes straight to HBM.  A workgroup walks B / gridDim.x samples, so a row that is hot in the
// batch is hot in every workgroup and reaches HBM once per workgroup instead of once per occurrence.
// (explicit LDS address space: through generic pointers the compiler falls back to flat atomics with a run-time
// "is it LDS?" test, which this toolchain then fails to encode)
typedef __attribute__((address_space(3))) int emb_lds_int;
typedef __attribute__((address_space(3))) unsigned long long emb_lds_u64;
struct EmbCache {
  emb_lds_int* tag;      // [NSLOT] id or -1
  emb_lds_u64* acc;      // [NSLOT, D]
  int nslot;             // power of two
};

// cache lookup of one id by the lane group's lane 0: returns the slot's owner after trying to claim an empty slot
__device__ __forceinline__ int emb_cache_claim(const EmbCache& c, unsigned int slot, int id) {
  int tag = c.tag[slot];
  if (tag == -1) {
    int expected = -1;
    __hip_atomic_compare_exchange_strong(c.tag + slot, &expected, id, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    tag = expected == -1 ? id : expected;
  }
  return tag;
}

__device__ __forceinline__ unsigned int emb_cache_slot(const EmbCache& c, int id) {
  return ((unsigned int)id * 2654435761u >> 7) & (unsigned int)(c.nslot - 1);
}

// An id with one occurrence in the batch needs no accumulator: its row is updated in place by the lane that holds it
// (nobody else reads that row in this step), with the same 2^-44 rounding as the accumulate path so that both give the
// same bits.
__device__ __forceinline__ void emb_apply_single(const EmbTrainArgs& a, int id, int l, float row, float g) {
  const long long q = emb_fix(g);
  a.emb[(long long)id * a.D + l] = row - a.lr * (float)((double)q * EMB_FIX_INV);
  if (l == 0) a.mark[emb_pidx(a, id)] = 0u;
}

__device__ __forceinline__ void emb_add(const EmbTrainArgs& a, const EmbCache& c, int id, unsigned int slot, int tag, int l, float g) {
  const long long q = emb_fix(g);
  if (q == 0) return;
  if (tag == id) { if (!(a.dbg & 4)) __hip_atomic_fetch_add(c.acc + (size_t)slot * a.D + l, (unsigned long long)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  else if (!(a.dbg & 2)) atomicAdd(reinterpret_cast<unsigned long long*>(a.accum + (long long)a.rank[emb_pidx(a, id)] * a.D + l), (unsigned long long)q);
}

constexpr int EMB_PASS = 4;               // rows a lane group has in flight
constexpr int EMB_GRAD_THREADS = 1024;   // 16 wavefronts share one cache, one workgroup per CU

// MODE: 0 = YouTube mean pooling, 1 = DIN cosine, 2 = DIN euclid — compile-time, like the attention kernels: run-time mode
// branches inside the unrolled stages put a branch (and a wait) around every shuffle.
// CACHE: false = no LDS staging (vocabularies much larger than the batch: almost every lookup would miss, and the
// claim + broadcast per slot is pure overhead)
template <int GS, int MODE, bool CACHE>
__global__ __launch_bounds__(EMB_GRAD_THREADS) void emb_grad_kernel(EmbTrainArgs a, int nslot) {
  extern __shared__ __attribute__((aligned(16))) unsigned char emb_smem[];
  EmbCache c;
  c.nslot = nslot;
  c.acc = (emb_lds_u64*)emb_smem;
  c.tag = (emb_lds_int*)(emb_smem + (size_t)nslot * a.D * sizeof(long long));
  if (CACHE) {
    for (int i = threadIdx.x; i < nslot * a.D; i += EMB_GRAD_THREADS) c.acc[i] = 0;
    for (int i = threadIdx.x; i < nslot; i += EMB_GRAD_THREADS) c.tag[i] = -1;
    __syncthreads();
  }
  constexpr int NG = 64 / GS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l = lane % GS, grp = lane / GS;
  const int D = a.D, T = a.T;
  const bool act = l < D;
  constexpr bool din = MODE != 0, cosine = MODE == 1;
  // slots a lane group has in flight: the mean-pooling mode has no per-slot arithmetic to hide the id -> row latency
  // behind (and needs few registers), the attention modes are issue-bound
  constexpr int PASS = MODE == 0 ? 8 : EMB_PASS;
  const float invT = 1.0f / (float)T;
  constexpr int WPB = EMB_GRAD_THREADS / 64;
  for (int b = blockIdx.x * WPB + wave; b < ((a.dbg & 8) ? 0 : a.B); b += gridDim.x * WPB) {
    const long long gr = a.st->batch_idx * (long long)a.B + b;
    if (gr >= a.src.rows) continue;                                 // padded row: no ids (wave-uniform)
    const int item = a.src.item_ids[gr];
    const bool item_ok = item >= 0 && item < a.V;
    const float v = (act && item_ok) ? a.emb[(long long)item * D + l] : 0.f;
    const float dp = act ? a.dpv[(size_t)b * a.ldp + l] : 0.f;
    const float nv = sqrtf(group_sum<GS>(v * v));
    const float inv_nv = nv > 0.f ? 1.0f / nv : 0.f;
    float dv = 0.f;
    // Memory chain per sample: ids + gates (one coalesced load each) -> all rows of a pass -> arithmetic; a pass is
    // PASS slots per lane group, their row loads are issued back to back before any of them is used.
    for (int tb = 0; tb < T; tb += 64) {
      // One group per wavefront (D > 32): the ids and their marks of 64 slots arrive with ONE coalesced load and ONE gather
      // (lane j fetches slot j's) and are handed out by v_readlane — as broadcast loads every slot would cost the
      // wavefront two dependent memory round trips of its own, and with 16 wavefronts per CU that latency is the kernel.
      int myid = -1;
      unsigned int mymark = 0u;
      if (NG == 1) {
        const int tl = tb + lane;
        const int id = tl < T ? a.src.ub_ids[gr * T + tl] : -1;
        myid = (id >= 0 && id < a.V) ? id : -1;
        mymark = myid >= 0 ? a.mark[emb_pidx(a, myid)] : 0u;
      }
      auto load_pass = [&](int pb, int (&ids)[PASS], float (&xs)[PASS], unsigned int (&mk)[PASS]) {
#pragma unroll
        for (int k = 0; k < PASS; ++k) {
          const int sl = pb + k * NG + grp;                         // slot within this 64-chunk
          const bool in = sl < 64 && tb + sl < T;
          if (NG == 1) {
            const int sel = __builtin_amdgcn_readfirstlane((pb + k) & 63);   // (uniform by construction; makes it an SGPR)
            ids[k] = in ? __builtin_amdgcn_readlane(myid, sel) : -1;
            mk[k] = in ? (unsigned int)__builtin_amdgcn_readlane((int)mymark, sel) : 0u;
          } else {
            const int id = in ? a.src.ub_ids[gr * T + tb + sl] : -1;   // one address per lane group: a broadcast load
            ids[k] = (id >= 0 && id < a.V) ? id : -1;
            mk[k] = ids[k] >= 0 ? a.mark[emb_pidx(a, ids[k])] : 0u;   // != 0: this pair is the id's only occurrence
          }
          xs[k] = (act && ids[k] >= 0 && !(a.dbg & 64)) ? a.emb[(long long)ids[k] * D + l] : 0.f;
        }
      };
      for (int pb = 0; pb < 64 && tb + pb < T; pb += PASS * NG) {
        int ids[PASS];
        float xs[PASS];
        unsigned int mk[PASS];
        load_pass(pb, ids, xs, mk);     // (requesting the next pass here as well was measured: no gain, the kernel is issue-bound)
        // every stage runs over all slots of the pass before the next one starts, so the LDS-crossbar shuffles and the
        // cache lookups of different slots are in flight together instead of one dependent chain per slot
        float g[PASS], att[PASS], s0[PASS], s1[PASS], s2[PASS];
        int tag[PASS];
#pragma unroll
        for (int k = 0; k < PASS; ++k) {
          const int sl = pb + k * NG + grp;
          const bool in = sl < 64 && tb + sl < T;
          g[k] = (din && in) ? a.gate[(size_t)b * T + tb + sl] : 1.0f;
          att[k] = (din && in) ? a.att0[tb + sl] : 0.f;
          tag[k] = (CACHE && l == 0 && ids[k] >= 0 && mk[k] == 0u && !(a.dbg & 32)) ? emb_cache_claim(c, emb_cache_slot(c, ids[k]), ids[k]) : -1;
          const float df = xs[k] - v;
          s0[k] = dp * xs[k];
          s1[k] = cosine ? xs[k] * xs[k] : (act ? df * df : 0.f);
          s2[k] = xs[k] * v;
        }
        if (din) {
#pragma unroll
          for (int k = 0; k < PASS; ++k) {                      // DPP sums (VALU): the LDS crossbar is the scarce unit here
            s0[k] = group_sum<GS>(s0[k]);
            s1[k] = group_sum<GS>(s1[k]);
            if (cosine) s2[k] = group_sum<GS>(s2[k]);
          }
        }
#pragma unroll
        for (int k = 0; k < PASS; ++k) tag[k] = CACHE ? emb_group_sum_int<GS>(l == 0 ? tag[k] + 1 : 0) - 1 : -1;   // broadcast of the group's lane 0
#pragma unroll
        for (int k = 0; k < PASS; ++k) {
          const int id = ids[k];
          if (id < 0 || (a.dbg & 16)) continue;                     // (group-uniform)
          const float x = xs[k];
          float dx;
          if (din) {
            dx = g[k] * invT * dp;
            const float q = s0[k] * invT * g[k] * (1.0f - g[k]) * att[k];
            if (cosine) {
              const float sxx = s1[k], sxy = s2[k];
              // reciprocals by v_rcp_f32 / v_rsq_f32 (1 ulp): the kernel is VALU-bound and an IEEE division is ~10 instructions
              const float inx = sxx > 0.f ? __frsqrt_rn(sxx) : 0.f, nx = sxx * inx;
              const float iden = __frcp_rn(nx * nv + 1e-8f);
              const float h = 0.5f * q * iden, sc = sxy * iden;
              dx += h * (v - sc * nv * inx * x);
              dv += h * (x - sc * nx * inv_nv * v);
            } else {
              const float df = x - v;
              const float ir = s1[k] > 0.f ? __frsqrt_rn(s1[k]) : 0.f;
              dx -= q * df * ir;
              dv += q * df * ir;
            }
          } else {
            dx = invT * dp;
          }
          if (act) {
            if (mk[k]) emb_apply_single(a, id, l, x, dx);
            else emb_add(a, c, id, emb_cache_slot(c, id), tag[k], l, dx);
          }
        }
      }
    }
    // candidate item: h0's item segment + the attention terms of every slot (sum over the lane groups)
#pragma unroll
    for (int o = GS; o < 64; o <<= 1) dv += __shfl_xor(dv, o, 64);
    if (grp == 0 && item_ok) {
      const bool single = a.mark[emb_pidx(a, item)] != 0u;
      const unsigned int slot = emb_cache_slot(c, item);
      int tg = (CACHE && l == 0 && !single) ? emb_cache_claim(c, slot, item) : -1;
      if (CACHE) tg = emb_group_sum_int<GS>(l == 0 ? tg + 1 : 0) - 1;
      if (act) {
        const float gsum = dv + a.dpv[(size_t)b * a.ldp + D + l];
        if (single) emb_apply_single(a, item, l, v, gsum);
        else emb_add(a, c, item, slot, tg, l, gsum);
      }
    }
  }
  if (!CACHE) return;
  __syncthreads();
  // flush the cached rows: one HBM atomic per (row, component) per workgroup
  for (int i = threadIdx.x; i < nslot * D; i += EMB_GRAD_THREADS) {
    const int tag = c.tag[i / D];
    const long long q = (long long)c.acc[i];
    if (tag >= 0 && q && !(a.dbg & 1)) atomicAdd(reinterpret_cast<unsigned long long*>(a.accum + (long long)a.rank[emb_pidx(a, tag)] * D + i % D), (unsigned long long)q);
  }
}

// =====================================================================================================================
// Round 3: the update WITHOUT contended atomics -- a per-batch sparse PLAN + an id-major accumulation.
//
// model.Train walks the resident rows in the same fixed batches epoch after epoch (model/model.go:96-211: no shuffle), so
// which (sample, slot) pairs of a batch touch which embedding row never changes.  That structure is computed ONCE per
// (dataset, batch size, vocabulary, world) and kept in HBM (12 B per pair -- 160 MB for bench.py's 262 144 rows, of 288 GB):
//     slots of batch k   = the distinct ids it touches, in ascending (owner-major) id order      slot_id, slot_off
//     pairs of batch k   = its (sample b, slot t) pairs sorted by slot                            pair = b << 12 | t, pslot, pid
// (emb_plan.hip: per batch a stable radix sort of the keys by owner-major row, a head-flag pass, one prefix sum and a fill --
// no atomics, byte-identical from build to build; once per dataset.)  A step then needs neither emb_mark, nor the rank scans, nor the LDS cache, nor emb_apply:
//   emb_coef   (DIN; one wavefront per sample, the attention kernels' layout) per pair the three scalars of
//                  dx_t = alpha dp + beta v + gamma x_t        (alpha = g / T;  cosine: beta = h, gamma = -h s |v| / |x|;
//                                                                euclid: beta = q / r, gamma = -q / r)
//              and per sample the item-row gradient gsum = dvh + sum_t (delta_t x_t) + (sum_t eps_t) v
//   emb_slot   id-major: a lane group walks EMB_SEG consecutive pairs of the sorted list, adds dx (2^-44 fixed point, so the
//              sum does not depend on the order the plan listed the pairs in) and, where a slot's run ends inside its
//              segment, writes E[id] -= lr * sum itself: ONE writer per row, no atomic, no second pass.  Runs that cross
//              segment borders are merged through LDS inside the 1024-thread workgroup (2048 pairs); only a run that
//              crosses a WORKGROUP border -- the Zipf head: an id with 30 000 occurrences in the batch is 15 of them --
//              goes through a 64-bit atomic add, and emb_span_apply finishes those rows.
// Same math as emb_grad_kernel and the same 2^-44 integer sums (bit-reproducible from run to run, and the exchange stays an
// integer sum); the per-pair expression is associated differently (alpha dp + beta v + gamma x), so the two paths agree to
// float32 rounding, not bit for bit.  emb_grad_kernel stays for embedding widths the attention layout does not cover (DIN
// with D not in {4, 8, 16, 32, 64}) and as the A/B reference (GOCTR_EMB_PLAN=0).

struct EmbPlanView {
  const int* pair;               // [pairs of all batches]  b << 12 | t   (t == T: the candidate item)
  const int* pslot;              // slot of the pair inside its batch
  const int* pid;                // embedding row of the pair
  const long long* pair_off;     // [nb + 1]
  const int* slot_id;            // [slots of all batches]
  const unsigned int* slot_off;  // per batch n_slots + 1 entries (relative to the batch's pairs): batch k's start at slot_base[k] + k
  const long long* slot_base;    // [nb + 1]
};
constexpr int EMB_PAIR_TBITS = 12;                 // t < 4096, b < 2^19
constexpr int EMB_SEG = 16;                        // pairs per lane group (one component per lane): all of them in flight at once
constexpr int EMB_SLOT_THREADS = 256;              // 16 lane groups at D = 16: 512 pairs per workgroup, several workgroups per CU

// ---- plan build (once per dataset): emb_plan.hip (a stable sort of each batch's keys + one flag / scan / fill pass)

// ---- per step, DIN: the per-pair coefficients (sample-major; the layout of attn_bwd_kernel)
struct EmbCoefArgs {
  RowSource src; const StepState* st;
  int B, T, D;
  const float* dpv; int ldp;          // [B, ldp]: 0..D-1 d cost / d pooled, D..2D-1 d cost / d item segment of h0
  const float* gate; const float* att0;
  float* dx;                          // [B, T, D]  the row gradient of every (sample, slot) pair: alpha dp + beta v + gamma x_t
  float* gsum;                        // [B, D]  gradient of the candidate item's row
  // attn_bwd_kernel's job rides along (the rows are gathered and dp . x is formed here anyway): the per-sample terms of the
  // att0 gradient, dgs [B, Tp] = (dp . x_t / T) g (1 - g) w_t  (null: not wanted)
  const float* wgt; float* partial; int Tp;
};
// MODE 1 = cosine, 2 = euclid.  VEC = 4, LPR = D / 4 lanes per row.
template <int LPR, int MODE>
__global__ __launch_bounds__(256) void emb_coef_kernel(EmbCoefArgs a) {
  constexpr int VEC = 4, RPP = 64 / LPR, NPB = LPR < 4 ? LPR : 4, SLOTS = NPB * RPP;
  constexpr bool cosine = MODE == 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x * 4 + wave;
  if (b >= a.B) return;
  const RowSource& s = a.src;
  const int dl = lane % LPR, rl = lane / LPR, d0 = dl * VEC;
  const int D = a.D, T = a.T;
  const long long gr = a.st->batch_idx * (long long)a.B + b;
  const bool valid = gr < s.rows;
  float dpt[VEC], vv[VEC], dvh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { dpt[e] = a.dpv[(size_t)b * a.ldp + d0 + e]; dvh[e] = a.dpv[(size_t)b * a.ldp + D + d0 + e]; }
  // (the first block's slot ids before the candidate row: both hang on the sample's index alone -- one latency less in the chain)
  int myid_first = -1;
  if (valid && lane < SLOTS && lane < T) myid_first = s.ub_ids[gr * T + lane];
  const int item = valid ? s.item_ids[gr] : -1;
  const bool item_ok = item >= 0 && item < s.V;
  load_row_nn<VEC>(s.emb + (long long)(item_ok ? item : s.V) * D, d0, D, true, vv);
  float syy = 0.f;
#pragma unroll
  for (int e = 0; e < VEC; ++e) syy += vv[e] * vv[e];
  const float nv = sqrtf(group_sum<LPR>(syy));
  const float inv_nv = nv > 0.f ? 1.0f / nv : 0.f;
  const float invT = 1.0f / (float)T;
  float dsum[VEC] = {0.f, 0.f, 0.f, 0.f};
  float esum = 0.f;
  if (a.partial)
    for (int t = T + lane; t < a.Tp; t += 64) a.partial[(size_t)b * a.Tp + t] = 0.f;
  for (int tb = 0; tb < T; tb += SLOTS) {
    int myid = myid_first;
    if (tb > 0) {
      myid = -1;
      if (valid && lane < SLOTS && tb + lane < T) myid = s.ub_ids[gr * T + tb + lane];
    }
    float gl = 0.f, al = 0.f;
    if (lane < SLOTS && tb + lane < T) { gl = a.gate[(size_t)b * T + tb + lane]; al = a.att0[tb + lane]; }
    float x[NPB][VEC];
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
      const int t = tb + p * RPP + rl;
      const int id = __shfl(myid, p * RPP + rl, 64);
      load_row_nn<VEC>(s.emb + (long long)((t < T && id >= 0 && id < s.V) ? id : s.V) * D, d0, D, true, x[p]);
    }
    const int src = (lane % RPP) * LPR, pw = lane / RPP;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;      // of slot tb + lane: dp . x, |x|^2 (euclid: |x - v|^2), x . v
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
      float u0 = 0.f, u1 = 0.f, u2 = 0.f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        u0 += dpt[e] * x[p][e];
        if (cosine) { u1 += x[p][e] * x[p][e]; u2 += x[p][e] * vv[e]; }
        else { const float df = x[p][e] - vv[e]; u1 += df * df; }
      }
      u0 = group_sum<LPR>(u0); u1 = group_sum<LPR>(u1);
      if (cosine) u2 = group_sum<LPR>(u2);
      const float v0 = __shfl(u0, src, 64), v1 = __shfl(u1, src, 64), v2 = cosine ? __shfl(u2, src, 64) : 0.f;
      if (pw == p) { s0 = v0; s1 = v1; s2 = v2; }
    }
    // the slot's coefficients, once, in its own lane (the expressions of emb_grad_kernel)
    const bool live = lane < SLOTS && tb + lane < T && myid >= 0 && myid < s.V;
    float alpha = 0.f, beta = 0.f, gamma = 0.f, delta = 0.f, eps = 0.f;
    if (live) {
      alpha = gl * invT;
      const float q = s0 * invT * gl * (1.0f - gl) * al;
      if (cosine) {
        const float inx = s1 > 0.f ? __frsqrt_rn(s1) : 0.f, nx = s1 * inx;
        const float iden = __frcp_rn(nx * nv + 1e-8f);
        const float h = 0.5f * q * iden, sc = s2 * iden;
        beta = h; gamma = -(h * (sc * nv * inx));
        delta = h; eps = -(h * (sc * nx * inv_nv));
      } else {
        const float ir = s1 > 0.f ? __frsqrt_rn(s1) : 0.f;
        beta = q * ir; gamma = -(q * ir);
        delta = q * ir; eps = -(q * ir);
      }
    }
    if (a.partial && lane < SLOTS && tb + lane < T)
      a.partial[(size_t)b * a.Tp + tb + lane] = (s0 * invT) * (gl * (1.0f - gl)) * a.wgt[(size_t)b * T + tb + lane];
    esum += eps;
    // The pair's whole row gradient leaves here, where dp, v and x_t are in registers: 64 contiguous bytes per pair, a pass
    // of 16 rows = 1 KiB per store instruction.  (The first version left three scalars per pair and emb_slot re-fetched dp, v
    // and x for every pair: 73 MB of 64-byte requests past the L2s, 31 us, SQ_WAIT_ANY 68 % -- the id-major side sees the
    // samples in random order.)
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
      const int sl = p * RPP + rl, t = tb + sl;
      const float d = __shfl(delta, sl, 64);
      const float pa = __shfl(alpha, sl, 64), pb = __shfl(beta, sl, 64), pg = __shfl(gamma, sl, 64);
      typedef float v4 __attribute__((ext_vector_type(4)));
      v4 o;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        dsum[e] += d * x[p][e];
        o[e] = pa * dpt[e] + pb * vv[e] + pg * x[p][e];
      }
      if (t < T) *reinterpret_cast<v4*>(a.dx + ((size_t)b * T + t) * D + d0) = o;
    }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) esum += __shfl_xor(esum, o, 64);
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const float dv = cross_row_sum<LPR>(dsum[e]) + esum * vv[e];
    if (rl == 0) a.gsum[(size_t)b * D + d0 + e] = dv + dvh[e];
  }
}

// ---- per step: id-major accumulation over the plan
struct EmbSlotArgs {
  EmbPlanView plan; const StepState* st;
  int B, T, D;
  const float* dpv; int ldp;
  const float* dx; const float* gsum;          // DIN (MODE != 0): per-pair row gradients [B, T, D] and the item rows' [B, D] (emb_coef)
  float* emb; long long* accum; float lr;
};

// MODE: 0 mean pooling (dx = dp / T), otherwise DIN (dx = alpha dp + beta v + gamma x).  DIRECT: single GPU -- finished rows are
// written to the table; false: every slot's sum goes to accum (the send buffer of the data-parallel exchange).
// A lane group = GS lanes x VEC components per lane (GS * VEC >= D).  VEC = 4: four lanes carry a 16-wide row with one
// 16-byte load each, so a wavefront instruction serves 16 pairs (64 / GS) instead of 4 -- the kernel is ISSUE-bound at cfg3
// (the per-pair bookkeeping: three hand-offs, address arithmetic, run tracking, was paid on every one of 16 lanes per pair).
// VEC = 1 is the generic layout (any D <= 64, any alignment).
template <int GS, int VEC>
struct EmbSlotGeo {
  static constexpr int SEG = VEC == 1 ? EMB_SEG : (GS * 4 < 16 ? 16 : GS * 4);      // pairs per lane group
  static constexpr int GPB = EMB_SLOT_THREADS / GS;                                 // lane groups per workgroup
  static constexpr long long WGP = (long long)GPB * SEG;                            // pairs per workgroup
};

template <int GS, int VEC, int MODE, bool DIRECT>
__global__ __launch_bounds__(EMB_SLOT_THREADS) void emb_slot_kernel(EmbSlotArgs a) {
  typedef float fv __attribute__((ext_vector_type(VEC)));
  constexpr int SEG = EmbSlotGeo<GS, VEC>::SEG, GPB = EmbSlotGeo<GS, VEC>::GPB, DW = GS * VEC;
  __shared__ long long entL[GPB][DW], entR[GPB][DW];
  __shared__ int slotL[GPB], slotR[GPB], bothL[GPB], idL[GPB], idR[GPB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NG = 64 / GS;
  const int l = lane % GS, grp = lane / GS, gib = wave * NG + grp;
  const int D = a.D, T = a.T;
  const int c0 = l * VEC;                            // first component of this lane
  const bool act = c0 < D;                           // (D % VEC == 0: a lane is all in or all out)
  const long long k = a.st->batch_idx;
  const long long pbase = a.plan.pair_off[k], npairs = a.plan.pair_off[k + 1] - pbase;
  const long long wg_begin = (long long)blockIdx.x * GPB * SEG;
  if (wg_begin >= npairs) return;
  const long long wg_end = wg_begin + (long long)GPB * SEG < npairs ? wg_begin + (long long)GPB * SEG : npairs;
  const int* pair = a.plan.pair + pbase;
  const int* pslot = a.plan.pslot + pbase;
  const int* pid = a.plan.pid + pbase;
  const long long beg = wg_begin + (long long)gib * SEG;
  const long long end = beg + SEG < wg_end ? beg + SEG : wg_end;
  const float invT = 1.0f / (float)T;
  if (l == 0) { slotL[gib] = -1; slotR[gib] = -1; bothL[gib] = 0; }

  auto ldv = [&](const float* p) -> fv {            // VEC consecutive floats (16-byte load when VEC == 4)
    fv r;
    if (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(p); r[0] = t.x; r[1 % VEC] = t.y; r[2 % VEC] = t.z; r[3 % VEC] = t.w; }
    else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) r[e] = p[e];
    }
    return r;
  };
  // a slot whose whole run was summed here: this lane group is the only writer of its row.  x = the row's value, loaded
  // together with the run's first pair (a load issued HERE would put one memory latency between every two runs: at cfg4
  // nearly every pair is a run of its own and the kernel took 100 us that way)
  auto finish = [&](int id, const fv& x, const long long (&q)[VEC]) {
    if (!act || !DIRECT) return;
    bool any = false;
    fv nv;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { any = any || q[e] != 0; nv[e] = x[e] - a.lr * (float)((double)q[e] * EMB_FIX_INV); }
    if (!any) return;
    float* p = a.emb + (long long)id * D + c0;
    if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(nv[0], nv[1 % VEC], nv[2 % VEC], nv[3 % VEC]);
    else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) p[e] = nv[e];
    }
  };
  if (beg < end) {
    const int prev_slot = beg > 0 ? pslot[beg - 1] : -1;
    const int next_slot = end < npairs ? pslot[end] : -1;
    // the segment's pairs: lane j of the group fetches pair j (and j + GS, ...): coalesced, handed round by shuffle
    constexpr int NL = SEG / GS > 0 ? SEG / GS : 1;
    int c_[NL], s_[NL], i_[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const long long i = beg + u * GS + l;
      const bool in = (GS >= SEG ? l < SEG : true) && i < end;
      c_[u] = in ? pair[i] : 0; s_[u] = in ? pslot[i] : -1; i_[u] = in ? pid[i] : 0;
    }
    int cur = -1, cur_id = 0, cur_prev = -2;          // cur_prev: slot of the last pair of the previous block of loads
    fv cur_x;
#pragma unroll
    for (int e = 0; e < VEC; ++e) cur_x[e] = 0.f;
    bool cur_ol = false;
    long long acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0;
    auto close = [&](bool open_right) {
      if (cur < 0) return;
      if (!cur_ol && !open_right) {
        if (DIRECT) finish(cur_id, cur_x, acc);
        else if (act) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) a.accum[(long long)cur * D + c0 + e] = acc[e];
        }
      } else if (cur_ol) {
        if (act) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) entL[gib][c0 + e] = acc[e];
        }
        if (l == 0) { slotL[gib] = cur; bothL[gib] = open_right ? 1 : 0; idL[gib] = cur_id; }
      } else {
        if (act) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) entR[gib][c0 + e] = acc[e];
        }
        if (l == 0) { slotR[gib] = cur; idR[gib] = cur_id; }
      }
    };
    // pairs whose rows are requested before the first of them is used (8 and 16 in flight were measured: no gain at cfg3, a
    // loss at cfg4 -- 78 -> 92 us with four components per lane, and the one-component layout spilled)
    constexpr int UNR = 4;
    constexpr int PER = GS < SEG ? GS : SEG;          // pairs one register set (c_[ub], s_[ub], i_[ub]) holds
    constexpr int UN = UNR < PER ? UNR : PER;
    static_assert(PER % UN == 0, "segment layout");
#pragma unroll
    for (int ub = 0; ub < NL; ++ub) {
      for (int jj = 0; jj < PER; jj += UN) {
        if (beg + ub * PER + jj >= end) break;
        int code[UN], sl[UN], id[UN];
        fv dp[UN], xx[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int srcl = grp * GS + jj + u;
          code[u] = __shfl(c_[ub], srcl, 64);
          sl[u] = __shfl(s_[ub], srcl, 64);
          id[u] = __shfl(i_[ub], srcl, 64);
          if (beg + ub * PER + jj + u >= end) sl[u] = -1;
          const unsigned b = (unsigned)code[u] >> EMB_PAIR_TBITS, t = (unsigned)code[u] & ((1u << EMB_PAIR_TBITS) - 1u);
          const bool on = sl[u] >= 0 && act;
#pragma unroll
          for (int e = 0; e < VEC; ++e) { dp[u][e] = 0.f; xx[u][e] = 0.f; }
          // the row itself is only needed where a run starts (it is applied at the run's end)
          const int before = u > 0 ? sl[u - 1] : (jj == 0 && ub == 0 ? -2 : cur_prev);
          if (on && DIRECT && sl[u] != before) xx[u] = ldv(a.emb + (long long)id[u] * D + c0);
          if (on) {
            if (MODE == 0) dp[u] = ldv(a.dpv + (b * (unsigned)a.ldp + ((int)t < T ? 0u : (unsigned)D) + (unsigned)c0));
            else dp[u] = (int)t < T ? ldv(a.dx + ((size_t)(b * (unsigned)T + t) * (unsigned)D + (unsigned)c0)) : ldv(a.gsum + (b * (unsigned)D + (unsigned)c0));
          }
        }
        cur_prev = sl[UN - 1] >= 0 ? sl[UN - 1] : cur_prev;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          if (sl[u] < 0) continue;                              // (group-uniform)
          if (sl[u] != cur) {
            close(false);
            cur = sl[u]; cur_id = id[u]; cur_x = xx[u];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = 0;
            cur_ol = (ub * PER + jj + u == 0) && cur == prev_slot;
          }
          const int t = code[u] & ((1 << EMB_PAIR_TBITS) - 1);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float dx = (MODE == 0 && t < T) ? invT * dp[u][e] : dp[u][e];
            acc[e] += emb_fix(dx);
          }
        }
      }
    }
    close(cur >= 0 && cur == next_slot);
  }
  __syncthreads();
  // runs that cross segment borders inside this workgroup: the entry that opens a chain sums it
  //   R(g) [closed left, open right] -> L(g+1) [open left] -> while that one is also open right: L(g+2) ...
  //   L(0) opens the chain that arrives from the previous workgroup
  const int ngroups = (int)((wg_end - wg_begin + SEG - 1) / SEG);
  if (gib < ngroups) {
    for (int which = 0; which < 2; ++which) {
      // which 0: this group's R entry; which 1: group 0's L entry (arrives from the previous workgroup)
      if (which == 1 && gib != 0) break;
      const int s0 = which == 0 ? slotR[gib] : slotL[0];
      if (s0 < 0) continue;
      long long tot[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) tot[e] = which == 0 ? entR[gib][act ? c0 + e : 0] : entL[0][act ? c0 + e : 0];
      const int id0 = which == 0 ? idR[gib] : idL[0];
      bool closed = which == 1 && !bothL[0];
      const bool from_prev = which == 1;
      if (!(which == 1 && !bothL[0])) {
        for (int j = gib + 1; j < ngroups; ++j) {
          if (slotL[j] != s0) break;                          // (cannot happen for a consistent plan; ends the chain)
#pragma unroll
          for (int e = 0; e < VEC; ++e) tot[e] += entL[j][act ? c0 + e : 0];
          if (!bothL[j]) { closed = true; break; }
        }
      }
      const bool whole = closed && !from_prev;                // the run began and ended inside this workgroup
      if (!act) continue;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        if (whole) {
          if (DIRECT) { if (tot[e]) a.emb[(long long)id0 * D + c0 + e] -= a.lr * (float)((double)tot[e] * EMB_FIX_INV); }
          else a.accum[(long long)s0 * D + c0 + e] = tot[e];
        } else if (tot[e]) {
          atomicAdd(reinterpret_cast<unsigned long long*>(a.accum + (long long)s0 * D + c0 + e), (unsigned long long)tot[e]);
        }
      }
    }
  }
}

// rows whose run crosses a workgroup border of emb_slot_kernel (DIRECT): their sums sit in accum; the FIRST border inside a
// run applies it and clears the accumulator.  One lane per (border, component); WGP = the slot kernel's pairs per workgroup.
__global__ __launch_bounds__(256) void emb_span_apply_kernel(EmbSlotArgs a, long long WGP) {
  const int D = a.D;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long w = gid / D + 1;                            // border index, >= 1
  const int l = (int)(gid % D);
  const long long k = a.st->batch_idx;
  const long long pbase = a.plan.pair_off[k], npairs = a.plan.pair_off[k + 1] - pbase;
  const long long pos = w * WGP;
  if (pos >= npairs) return;
  const int* pslot = a.plan.pslot + pbase;
  const int s = pslot[pos];
  if (pslot[pos - 1] != s) return;
  const long long sb = a.plan.slot_base[k];
  const unsigned int start = a.plan.slot_off[sb + k + s];
  if ((long long)(start / WGP) + 1 != w) return;              // an earlier border of the same run does it
  const int id = a.plan.slot_id[sb + s];
  const long long q = a.accum[(long long)s * D + l];
  a.accum[(long long)s * D + l] = 0;
  if (q) a.emb[(long long)id * D + l] -= a.lr * (float)((double)q * EMB_FIX_INV);
}

// data parallel: the exchange (ctr.hip: launch_emb_exchange) wants the batch's slot -> id list and slot count in fixed buffers
__global__ void emb_plan_select_kernel(EmbPlanView plan, const StepState* st, int* slot_id_out, unsigned long long* n_slots_out) {
  const long long k = st->batch_idx;
  const long long sb = plan.slot_base[k], n = plan.slot_base[k + 1] - sb;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) slot_id_out[i] = plan.slot_id[sb + i];
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_slots_out = (unsigned long long)n;
}

// Sink of the rank scan (scan.h): rank of every id, slot -> id list of the touched ones (ascending ids), and the marks
// are cleared for the next step in the same pass.
struct EmbMultiMap {
  __device__ __forceinline__ unsigned int operator()(unsigned int v) const { return v == EMB_MULTI ? 1u : 0u; }
};
struct EmbRankSink {
  unsigned int* mark; unsigned int* rank; int* slot_id;
  int W; long long Vw; long long base;    // the scanned range starts at permuted index `base` (owner-side scans cover one bucket)
  __device__ __forceinline__ void operator()(long long i, unsigned int m, unsigned int r) const {
    if (m == EMB_MULTI) {     // (a single keeps its mark until emb_grad has applied it)
      const long long pi = base + i;
      const long long id = W == 1 ? pi : (pi % Vw) * W + pi / Vw;
      rank[pi] = r; slot_id[r] = (int)id; mark[pi] = 0u;
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// Bucketed exchange of the sparse row gradients (data parallel, replicated table).  Owner of an
// id = id % W.  After the local rank scan a rank's rows for owner o are accum[off[o] .. off[o+1]) (ids ascending):
//   1. all-to-all of (ids, 64-bit fixed-point rows), exact counts;
//   2. the owner marks the received ids inside ITS bucket of the mark array, scans that bucket -> dense slots of the
//      unique ids, and adds every received row into red[slot] -- integer adds, exact and order-independent;
//   3. delta = lr * float(sum) (the expression emb_apply uses), ids + deltas all-gathered;
//   4. every rank subtracts every delta from its replica: the replicas stay bit-identical.
// Traffic is proportional to the ids the batch touches, not to the vocabulary.

// off[o] = first slot whose owner is >= o (slot_id is sorted by (owner, id)); off[W] = n
__global__ void emb_bucket_bounds_kernel(const int* slot_id, const unsigned long long* n_slots, int W, int* off, int* cnt) {
  const int o = threadIdx.x;
  const long long n = (long long)*n_slots;
  if (o > W) return;
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (slot_id[mid] % W >= o) hi = mid; else lo = mid + 1;
  }
  off[o] = (int)lo;
  __syncthreads();
  if (o < W) cnt[o] = off[o + 1] - off[o];
}

// owner side: mark the received ids (all of them fall into this rank's bucket)
__global__ void emb_recv_mark_kernel(const int* rids, long long n, int W, long long Vw, unsigned int* mark) {
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int id = rids[k];
  if (id < 0) return;                                          // (padding of a fixed-size bucket)
  mark[(long long)(id % W) * Vw + id / W] = EMB_MULTI;
}

// owner side: red[rank[pidx(id_k)]][l] += rows[k][l]
__global__ void emb_recv_accumulate_kernel(const int* rids, const long long* rows, long long n, int D, int W, long long Vw,
                                           const unsigned int* rank, long long* red) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * D) return;
  const long long k = i / D;
  const int l = (int)(i - k * D), id = rids[k];
  if (id < 0) return;
  const long long q = rows[i];
  if (q) atomicAdd(reinterpret_cast<unsigned long long*>(red + (long long)rank[(long long)(id % W) * Vw + id / W] * D + l), (unsigned long long)q);
}

// owner side: delta = lr * float(sum) (emb_apply's expression), red cleared behind
__global__ void emb_delta_kernel(long long* red, const unsigned long long* n_red, int D, float lr, float* delta) {
  const long long n = (long long)*n_red * D;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long q = red[i];
    red[i] = 0;
    delta[i] = lr * (float)((double)q * EMB_FIX_INV);
  }
}

__global__ void emb_count_to_i32_kernel(const unsigned long long* n, int* out) { *out = (int)*n; }

// every rank: E[id] -= delta for the gathered (id, delta) lists of all owners (ids are unique across the lists)
__global__ void emb_apply_gathered_kernel(float* emb, const int* ids, const float* delta, long long n, int D) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n * D; i += (long long)gridDim.x * 256) {
    const float d = delta[i];
    const int id = ids[i / D];
    if (id >= 0 && d != 0.f) emb[(long long)id * D + i % D] -= d;
  }
}

// E[id] -= lr * accum, accum cleared behind; grid-stride over (slot, component), n = the scan's total
__global__ void emb_apply_kernel(EmbTrainArgs a, const int* slot_id, const unsigned long long* n_slots) {
  const long long n = (long long)*n_slots * a.D;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long q = a.accum[i];
    if (q == 0) continue;
    a.accum[i] = 0;
    a.emb[(long long)slot_id[i / a.D] * a.D + i % a.D] -= a.lr * (float)((double)q * EMB_FIX_INV);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The exchange with FIXED-SIZE buckets (round 3): with a plan every batch's bucket sizes are known when the plan is built,
// so the bound S = the largest bucket any rank sends for any batch is exact (no overflow possible) and the owner-side bound
// is R = min(Vw, W S).  Buckets travel padded to S (ids -1, rows 0), the owners' (id, delta) lists padded to R: every
// transfer has a size the host knows in advance -- no counts all-gather, no read-back, nothing between the collectives
// but captured kernels.  Counts travel in-band (the -1 padding).

// bucket bounds of every batch of the plan (once, at plan build): off[k][o] = first slot of batch k whose owner is >= o
__global__ void emb_plan_buckets_kernel(EmbPlanView plan, long long nb, int W, int* off) {
  const long long k = blockIdx.x;
  const int o = threadIdx.x;
  if (k >= nb || o > W) return;
  const long long sb = plan.slot_base[k], n = plan.slot_base[k + 1] - sb;
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (plan.slot_id[sb + mid] % W >= o) hi = mid; else lo = mid + 1;
  }
  off[k * (W + 1) + o] = (int)lo;
}

// send buffers of the running batch: for owner o, S (id, row) entries -- its bucket, then padding; the accumulators are
// cleared behind (the next step's border atomics add into zeros)
__global__ void emb_pack_send_kernel(EmbPlanView plan, const StepState* st, const int* bucket_off, int W, int S, int D, long long* accum,
                                     int* send_ids, long long* send_rows) {
  const long long k = st->batch_idx;
  const long long sb = plan.slot_base[k];
  const int* off = bucket_off + k * (W + 1);
  const long long n = (long long)W * S * D;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long e = i / D;
    const int l = (int)(i - e * D), o = (int)(e / S), j = (int)(e - (long long)o * S);
    const int slot = off[o] + j;
    const bool live = slot < off[o + 1];
    long long q = 0;
    if (live) { q = accum[(long long)slot * D + l]; accum[(long long)slot * D + l] = 0; }
    send_rows[i] = q;
    if (l == 0) send_ids[e] = live ? plan.slot_id[sb + slot] : -1;
  }
}

// owner side, after emb_delta: the id list padded to R entries for the fixed-size all-gather
__global__ void emb_pad_ids_kernel(int* red_ids, const unsigned long long* n_red, int R) {
  const long long n = (long long)*n_red;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < R; i += (long long)gridDim.x * 256)
    if (i >= n) red_ids[i] = -1;
}

}  // namespace goctr
