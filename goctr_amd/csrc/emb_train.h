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
//                 ASCENDING ID ORDER (deterministic numbering, no hashing, no sort)
//   w0pv_t, gemm  dpv[B, 2D] = dz0 . W0[U : U+2D, :]^T   (MFMA; dz0 is what the chain kernel already stored)
//   emb_grad      one lane group per (sample, slot): row gradient -> 64-bit FIXED-POINT atomic adds into accum[rank[id]]
//   emb_apply     E[id] -= lr * accum ; clears accum and mark behind itself
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

__device__ __forceinline__ void emb_accumulate(long long* dst, float g) {
  const long long q = __double2ll_rn((double)g * EMB_FIX_SCALE);
  if (q) atomicAdd(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)q);
}

// One wavefront per sample, 64 / GS lane groups; a group owns slots t = grp, grp + NG, ...; lane l of a group owns
// embedding component l (D <= GS <= 64).
template <int GS>
__global__ __launch_bounds__(256) void emb_grad_kernel(EmbTrainArgs a) {
  constexpr int NG = 64 / GS;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const long long gr = a.st->batch_idx * (long long)a.B + b;
  if (gr >= a.src.rows) return;
  const int l = lane % GS, grp = lane / GS;
  const int D = a.D, T = a.T;
  const bool act = l < D;
  const bool din = a.kind == GOCTR_DIN, cosine = a.att == GOCTR_ATT_COSINE;
  const float invT = 1.0f / (float)T;
  const int item = a.src.item_ids[gr];
  const bool item_ok = item >= 0 && item < a.V;
  const float v = (act && item_ok) ? a.emb[(long long)item * D + l] : 0.f;
  const float dp = act ? a.dpv[(size_t)b * a.ldp + l] : 0.f;
  const float nv = sqrtf(emb_group_sum<GS>(v * v));
  float dv = 0.f;
  for (int t = grp; t < T; t += NG) {
    const int id = a.src.ub_ids[gr * T + t];
    if (id < 0 || id >= a.V) continue;                            // (group-uniform)
    const float x = act ? a.emb[(long long)id * D + l] : 0.f;
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
    if (act) emb_accumulate(a.accum + (long long)a.rank[id] * D + l, dx);
  }
  // candidate item: h0's item segment + the attention terms of every slot (sum over the lane groups)
#pragma unroll
  for (int o = GS; o < 64; o <<= 1) dv += __shfl_xor(dv, o, 64);
  if (grp == 0 && act && item_ok) emb_accumulate(a.accum + (long long)a.rank[item] * D + l, dv + a.dpv[(size_t)b * a.ldp + D + l]);
}

// One wavefront per 64 consecutive ids; the touched ones are processed one after the other with all lanes on the row.
__global__ __launch_bounds__(256) void emb_apply_kernel(EmbTrainArgs a) {
  const int lane = threadIdx.x & 63;
  const long long id0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (id0 >= a.V) return;
  const long long mine = id0 + lane;
  const unsigned int mk = mine < a.V ? a.mark[mine] : 0u;
  unsigned long long todo = __ballot(mk != 0u);
  if (!todo) return;
  const unsigned int rk = mk ? a.rank[mine] : 0u;
  if (mk) a.mark[mine] = 0u;
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const unsigned int u = __shfl(rk, src, 64);
    const long long id = id0 + src;
    for (int d = lane; d < a.D; d += 64) {
      long long* ap = a.accum + (long long)u * a.D + d;
      const float g = (float)((double)*ap * EMB_FIX_INV);
      *ap = 0;
      a.emb[id * a.D + d] -= a.lr * g;
    }
  }
}

}  // namespace goctr
