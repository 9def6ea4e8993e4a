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
//   emb_mark      mark[id] = 1 for every id the batch touches                         (B (T+1) ids)
//   scan x3       rank = exclusive prefix sum of mark over the vocabulary: the touched ids get dense slots 0..n-1 in
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
  unsigned int* mark;              // [V]
  const unsigned int* rank;        // [V]
  long long* accum;                // [min(V, B (T+1)), D]
  float lr;
};

__global__ void emb_mark_kernel(EmbTrainArgs a) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const int per = a.T + 1;
  if (p >= (long long)a.B * per) return;
  const int b = (int)(p / per), t = (int)(p % per);
  const long long gr = a.st->batch_idx * (long long)a.B + b;
  if (gr >= a.src.rows) return;                                   // padded row: no ids
  const int id = t < a.T ? a.src.ub_ids[gr * a.T + t] : a.src.item_ids[gr];
  if (id >= 0 && id < a.V) a.mark[id] = 1u;                       // idempotent: racing writers store the same value
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

__device__ __forceinline__ long long emb_fix(float g) { return __double2ll_rn((double)g * EMB_FIX_SCALE); }

// LDS staging of hot rows: item popularity is Zipfian, and global atomics on one row serialise (the same effect as in
// the dictionary build, corpus.hip).  Each workgroup owns a direct-mapped cache of NSLOT rows of 64-bit accumulators in
// LDS: the first id that hashes to a slot claims it; its later occurrences inside this workgroup are LDS atomics, every
// other id of that slot goes straight to HBM.  A workgroup walks B / gridDim.x samples, so a row that is hot in the
// batch is hot in every workgroup and reaches HBM once per workgroup instead of once per occurrence.
struct EmbCache {
  int* tag;          // [NSLOT] id or -1
  long long* acc;    // [NSLOT, D]
  int nslot;         // power of two
};

template <int GS>
__device__ __forceinline__ void emb_accumulate(const EmbTrainArgs& a, const EmbCache& c, int id, int l, bool act, float g) {
  const unsigned int slot = ((unsigned int)id * 2654435761u >> 7) & (unsigned int)(c.nslot - 1);
  int tag = 0;
  if (l == 0) {
    tag = c.tag[slot];
    if (tag == -1) {
      const int old = atomicCAS(&c.tag[slot], -1, id);
      tag = old == -1 ? id : old;
    }
  }
  tag = __shfl(tag, (threadIdx.x & 63) / GS * GS, 64);       // the group's lane 0
  if (!act) return;
  const long long q = emb_fix(g);
  if (q == 0) return;
  if (tag == id) atomicAdd(reinterpret_cast<unsigned long long*>(c.acc + (size_t)slot * a.D + l), (unsigned long long)q);
  else atomicAdd(reinterpret_cast<unsigned long long*>(a.accum + (long long)a.rank[id] * a.D + l), (unsigned long long)q);
}

// One wavefront per sample at a time, 64 / GS lane groups; a group owns slots t = grp, grp + NG, ...; lane l of a
// group owns embedding component l (D <= GS <= 64).  Workgroup w takes samples 16 w .. 16 w + 15, then strides by the grid.
constexpr int EMB_GRAD_THREADS = 1024;   // 16 wavefronts share one cache; two workgroups per CU hide the id -> row -> atomic latency chain

template <int GS>
__global__ __launch_bounds__(EMB_GRAD_THREADS) void emb_grad_kernel(EmbTrainArgs a, int nslot) {
  extern __shared__ __attribute__((aligned(16))) unsigned char emb_smem[];
  EmbCache c;
  c.nslot = nslot;
  c.acc = reinterpret_cast<long long*>(emb_smem);
  c.tag = reinterpret_cast<int*>(emb_smem + (size_t)nslot * a.D * sizeof(long long));
  for (int i = threadIdx.x; i < nslot * a.D; i += EMB_GRAD_THREADS) c.acc[i] = 0;
  for (int i = threadIdx.x; i < nslot; i += EMB_GRAD_THREADS) c.tag[i] = -1;
  __syncthreads();
  constexpr int NG = 64 / GS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l = lane % GS, grp = lane / GS;
  const int D = a.D, T = a.T;
  const bool act = l < D;
  const bool din = a.kind == GOCTR_DIN, cosine = a.att == GOCTR_ATT_COSINE;
  const float invT = 1.0f / (float)T;
  constexpr int WPB = EMB_GRAD_THREADS / 64;
  for (int b = blockIdx.x * WPB + wave; b < a.B; b += gridDim.x * WPB) {
    const long long gr = a.st->batch_idx * (long long)a.B + b;
    if (gr >= a.src.rows) continue;                                 // padded row: no ids (wave-uniform)
    const int item = a.src.item_ids[gr];
    const bool item_ok = item >= 0 && item < a.V;
    const float v = (act && item_ok) ? a.emb[(long long)item * D + l] : 0.f;
    const float dp = act ? a.dpv[(size_t)b * a.ldp + l] : 0.f;
    const float nv = sqrtf(emb_group_sum<GS>(v * v));
    float dv = 0.f;
    // software pipeline: the next slot's id and row are requested before this slot's arithmetic and atomics
    int id_n = grp < T ? a.src.ub_ids[gr * T + grp] : -1;
    float x_n = (act && id_n >= 0 && id_n < a.V) ? a.emb[(long long)id_n * D + l] : 0.f;
    for (int t = grp; t < T; t += NG) {
      const int id = id_n;
      const float x = x_n;
      id_n = t + NG < T ? a.src.ub_ids[gr * T + t + NG] : -1;
      x_n = (act && id_n >= 0 && id_n < a.V) ? a.emb[(long long)id_n * D + l] : 0.f;
      if (id < 0 || id >= a.V) continue;                            // (group-uniform)
      float dx;
      if (din) {
        const float g = a.gate[(size_t)b * T + t];
        dx = g * invT * dp;
        const float q = emb_group_sum<GS>(dp * x) * invT * g * (1.0f - g) * a.att0[t];
        if (cosine) {
          const float sxx = emb_group_sum<GS>(x * x), sxy = emb_group_sum<GS>(x * v);
          const float nx = sqrtf(sxx), den = nx * nv + 1e-8f;
          const float cx = nx > 0.f ? sxy * nv / (nx * den * den) : 0.f;
          const float cv = nv > 0.f ? sxy * nx / (nv * den * den) : 0.f;
          dx += q * 0.5f * (v / den - cx * x);
          dv += q * 0.5f * (x / den - cv * v);
        } else {
          const float df = x - v;
          const float r = sqrtf(emb_group_sum<GS>(act ? df * df : 0.f));
          if (r > 0.f) {
            dx -= q * df / r;
            dv += q * df / r;
          }
        }
      } else {
        dx = invT * dp;
      }
      emb_accumulate<GS>(a, c, id, l, act, dx);
    }
    // candidate item: h0's item segment + the attention terms of every slot (sum over the lane groups)
#pragma unroll
    for (int o = GS; o < 64; o <<= 1) dv += __shfl_xor(dv, o, 64);
    if (grp == 0 && item_ok) emb_accumulate<GS>(a, c, item, l, act, dv + (act ? a.dpv[(size_t)b * a.ldp + D + l] : 0.f));
  }
  __syncthreads();
  // flush the cached rows: one HBM atomic per (row, component) per workgroup
  for (int i = threadIdx.x; i < nslot * D; i += EMB_GRAD_THREADS) {
    const int tag = c.tag[i / D];
    const long long q = c.acc[i];
    if (tag >= 0 && q) atomicAdd(reinterpret_cast<unsigned long long*>(a.accum + (long long)a.rank[tag] * D + i % D), (unsigned long long)q);
  }
}

// Sink of the rank scan (scan.h): rank of every id, slot -> id list of the touched ones (ascending ids), and the marks
// are cleared for the next step in the same pass.
struct EmbRankSink {
  unsigned int* mark; unsigned int* rank; int* slot_id;
  __device__ __forceinline__ void operator()(long long id, unsigned int m, unsigned int r) const {
    if (m) { rank[id] = r; slot_id[r] = (int)id; mark[id] = 0u; }
  }
};

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
