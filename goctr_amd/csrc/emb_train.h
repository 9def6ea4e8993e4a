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
  int dbg;                         // timing experiments only (GOCTR_EMB_DBG): 1 skip the flush, 2 skip cache misses, 4 skip LDS adds
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
  mark[(long long)(id % W) * Vw + id / W] = EMB_MULTI;
}

// owner side: red[rank[pidx(id_k)]][l] += rows[k][l]
__global__ void emb_recv_accumulate_kernel(const int* rids, const long long* rows, long long n, int D, int W, long long Vw,
                                           const unsigned int* rank, long long* red) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * D) return;
  const long long k = i / D;
  const int l = (int)(i - k * D), id = rids[k];
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
    if (d != 0.f) emb[(long long)ids[i / D] * D + i % D] -= d;
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

}  // namespace goctr
