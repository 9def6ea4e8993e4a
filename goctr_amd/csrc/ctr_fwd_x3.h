// Forward-only DIN / YouTube-DNN layers 0-2 for the recommend / predict path on the 6-product bf16 split
// (scripts/ubench/bf16x3.hip): replaces ctr_fwd16_kernel (v_mfma_f32_16x16x4_f32) when a predict batch
// gives fewer than one 32-row workgroup per CU.  Reference path: model.Predict -> the forward-only graph of
// din.go:219-323 / dnn.go:162-184 (A0 = sigm(h0 W0), A1 = sigm(A0 W1), y = sigm(A1 W2), no dropout: quirk Q1).
//
// Why here first: a 16-row workgroup stages the whole of W0 and W1 for 16 rows of work, and with the f32-input MFMA that
// staging does not overlap with the multiplication (the instruction blocks its SIMD: ubench mfma_dma_overlap).  The bf16
// MFMA does overlap, so the LDS-DMA of the loader wavefronts finally runs under the multipliers' MFMAs.
//
// Every float32 value x is used as  hi + mid + lo  (three round-to-nearest-even bf16, each of what the previous left) and
// a product as  hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid)  with float32 accumulation (two accumulators per
// tile): measured MORE accurate than the f32 MFMA on this path's operands (2.8e-8 vs 1.3e-7 of sum |a b|).
//
// Operand images (HBM, built from the float32 weights by fx_build_images_kernel whenever they changed since the last
// predict): one block of 1 KiB per (32-deep k chunk, 16-column tile, plane) laid out [q][i][8 bf16] — lane (i, q) of
// v_mfma_f32_16x16x32_bf16 reads its 16 bytes at (q * 16 + i) * 16: conflict-free, and a straight LDS-DMA copy.
//   W0 image: [chunk c][tile t][plane p] ; k = 32 c + 8 q + j, n = 16 t + i
//   W1 image: [chunk cc][tile u][plane p] ; the chunks follow the H1 tiles the multiplying wavefronts own (4, 4, 4, rest):
//             chunk = a pair of that wavefront's tiles (ta, tb), k = 16 ta + 4 q + j (j < 4) or 16 tb + 4 q + j - 4 — exactly
//             the order in which the accumulators of layer 0 sit in the lanes, so A0 feeds layer 1 without a shuffle.
#pragma once
#include "ctr_chain.h"

namespace goctr {

typedef __bf16 fx_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 fx_bf2 __attribute__((ext_vector_type(2)));
typedef float fx_f2 __attribute__((ext_vector_type(2)));
constexpr int FX_BLOCK = 1024;     // bytes of one (chunk, tile, plane) block
constexpr int FX_MAXC0 = 8;        // Ip <= 240 (chain_ok) => at most 8 chunks of 32

inline int fx_w0_chunks(int Ip) { return (Ip + 31) / 32; }
inline int fx_wave_tiles(int nt0, int w) { int n = nt0 - 4 * w; return n < 0 ? 0 : (n > 4 ? 4 : n); }
inline int fx_w1_chunks(int H1p) {
  int n = 0;
  for (int w = 0; w < 4; ++w) n += (fx_wave_tiles(H1p / 16, w) + 1) / 2;
  return n;
}
inline size_t fx_w0_bytes(int Ip, int H1p) { return (size_t)fx_w0_chunks(Ip) * (H1p / 16) * 3 * FX_BLOCK; }
inline size_t fx_w1_bytes(int H1p, int H2p) { return (size_t)fx_w1_chunks(H1p) * (H2p / 16) * 3 * FX_BLOCK; }
inline size_t fx_lds_bytes(int H1p, int H2p) {
  const size_t chunk = (size_t)(H1p / 16) * 3 * FX_BLOCK, w1 = fx_w1_bytes(H1p, H2p);
  return chunk + (w1 > chunk ? w1 : chunk);
}

// x0, x1 -> packed (hi, mid, lo) pairs, low half = x0 (v_cvt_pk_bf16_f32, RNE)
__device__ __forceinline__ void fx_split3_pk(float x0, float x1, unsigned int& hi, unsigned int& mid, unsigned int& lo) {
  hi = __builtin_bit_cast(unsigned int, __builtin_convertvector(fx_f2{x0, x1}, fx_bf2));
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xFFFF0000u);
  mid = __builtin_bit_cast(unsigned int, __builtin_convertvector(fx_f2{r0, r1}, fx_bf2));
  const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xFFFF0000u);
  lo = __builtin_bit_cast(unsigned int, __builtin_convertvector(fx_f2{s0, s1}, fx_bf2));
}
typedef unsigned int fx_u4 __attribute__((ext_vector_type(4)));
// 8 float32 -> three bf8 operands
__device__ __forceinline__ void fx_split8(const float (&x)[8], fx_bf8& hi, fx_bf8& mid, fx_bf8& lo) {
  fx_u4 h, m, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned int a, b, c;
    fx_split3_pk(x[2 * e], x[2 * e + 1], a, b, c);
    h[e] = a; m[e] = b; l[e] = c;
  }
  hi = __builtin_bit_cast(fx_bf8, h); mid = __builtin_bit_cast(fx_bf8, m); lo = __builtin_bit_cast(fx_bf8, l);
}

// the six products of one k chunk into (hh, corr): smallest terms first inside corr
__device__ __forceinline__ void fx_mma6(const fx_bf8 (&w)[3], const fx_bf8 (&x)[3], chain_f4& hh, chain_f4& corr) {
  corr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], x[0], corr, 0, 0, 0);
  corr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[2], corr, 0, 0, 0);
  corr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[1], corr, 0, 0, 0);
  corr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[0], corr, 0, 0, 0);
  corr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[1], corr, 0, 0, 0);
  hh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[0], hh, 0, 0, 0);
}

// ---- operand images from the float32 weights (W: the model's flat padded buffer, W0 [Ip, H1p] then W1 [H1p, H2p] at off1)
__global__ void fx_build_images_kernel(const float* W, int Ip, int H1p, int H2p, int off1, unsigned short* w0x, unsigned short* w1x,
                                       int nc0, int ncc) {
  const int NT0 = H1p >> 4, NT1 = H2p >> 4;
  const long long n0 = (long long)nc0 * NT0 * 512, n1 = (long long)ncc * NT1 * 512;     // (q, i, j) triples per plane
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  float x = 0.f;
  unsigned short* dst = nullptr;
  if (idx < n0) {
    const int j = (int)(idx & 7), i = (int)(idx >> 3) & 15, q = (int)(idx >> 7) & 3;
    const int t = (int)((idx >> 9) % NT0), c = (int)((idx >> 9) / NT0);
    const int k = 32 * c + 8 * q + j, n = 16 * t + i;
    x = k < Ip ? W[(size_t)k * H1p + n] : 0.f;
    dst = w0x + ((size_t)(c * NT0 + t) * 3) * 512 + (q * 16 + i) * 8 + j;
  } else if (idx < n0 + n1) {
    const long long e = idx - n0;
    const int j = (int)(e & 7), i = (int)(e >> 3) & 15, q = (int)(e >> 7) & 3;
    const int u = (int)((e >> 9) % NT1), cc = (int)((e >> 9) / NT1);
    // chunk cc -> (wavefront w, its local chunk): wavefronts own 4, 4, 4, rest tiles
    int w = 0, left = cc;
    for (; w < 4; ++w) {
      int nt = NT0 - 4 * w; nt = nt < 0 ? 0 : (nt > 4 ? 4 : nt);
      const int nch = (nt + 1) / 2;
      if (left < nch) break;
      left -= nch;
    }
    int ntw = NT0 - 4 * w; ntw = ntw > 4 ? 4 : ntw;
    const int tl = 2 * left + (j >= 4 ? 1 : 0);                  // local tile of this k slot
    const int k = 16 * (4 * w + tl) + 4 * q + (j & 3);
    x = (tl < ntw && k < H1p) ? W[(size_t)off1 + (size_t)k * H2p + 16 * u + i] : 0.f;
    dst = w1x + ((size_t)(cc * NT1 + u) * 3) * 512 + (q * 16 + i) * 8 + j;
  } else {
    return;
  }
  const __bf16 hi = (__bf16)x;
  const float r1 = x - (float)hi;
  const __bf16 mid = (__bf16)r1;
  const __bf16 lo = (__bf16)(r1 - (float)mid);
  dst[0] = __builtin_bit_cast(unsigned short, hi);
  dst[512] = __builtin_bit_cast(unsigned short, mid);
  dst[1024] = __builtin_bit_cast(unsigned short, lo);
}

// ---- the kernel: 16 rows per workgroup; wavefronts 0-3 multiply (H1 tiles 4, 4, 4, rest), 4-7 stage by LDS-DMA
template <int NT1>
__global__ __launch_bounds__(512, 1) void ctr_fwd16_x3_kernel(ChainArgs a, const unsigned short* w0x, const unsigned short* w1x, int nc0,
                                                               int ncc) {
  typedef chain_f4 f4;
  extern __shared__ __attribute__((aligned(16))) unsigned char fx_smem[];
  const int NT0 = a.H1p >> 4;
  const int chunkB = NT0 * 3 * FX_BLOCK;                 // bytes of one W0 chunk (all tiles, 3 planes)
  unsigned char* const bufA = fx_smem;
  unsigned char* const bufB = fx_smem + chunkB;          // W1 lands from here on (bufB + what lies behind it)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int first = (nc0 & 1) ? 0 : 1;                   // the LAST W0 phase must sit in bufA (W1 is staged over bufB meanwhile)
  auto buf_of = [&](int ph) { return ((ph + first) & 1) ? bufB : bufA; };

  if (wave >= 4) {                                       // loader wavefronts: the barrier sequence mirrors the multipliers'
    ChainStager stg;
    const int lw = wave - 4;
    stg.begin(reinterpret_cast<const float*>(w0x), reinterpret_cast<float*>(buf_of(0)), chunkB >> 2, lw);
    stg.drain(lane);
    __syncthreads();
    for (int ph = 0; ph < nc0; ++ph) {
      if (ph + 1 < nc0)
        stg.begin(reinterpret_cast<const float*>(w0x + (size_t)(ph + 1) * (chunkB >> 1)), reinterpret_cast<float*>(buf_of(ph + 1)), chunkB >> 2, lw);
      else
        stg.begin(reinterpret_cast<const float*>(w1x), reinterpret_cast<float*>(bufB), (ncc * NT1 * 3 * FX_BLOCK) >> 2, lw);
      stg.drain(lane);
      __syncthreads();
    }
    __syncthreads();                                     // Z1 exchange barrier
    return;
  }

  const int i = lane & 15, q = lane >> 4;
  const int row = blockIdx.x * 16 + i;
  const bool vrow = row < a.B;
  const int rowc = vrow ? row : a.B - 1;
  const int t0 = 4 * wave;
  int ntl = NT0 - t0; ntl = ntl < 0 ? 0 : (ntl > 4 ? 4 : ntl);

  // layer-0 activations of this lane's row: 8 consecutive inputs per chunk -> three bf16 planes
  fx_bf8 hb[FX_MAXC0][3];
#pragma unroll
  for (int c = 0; c < FX_MAXC0; ++c) {
    if (c < nc0) {
      float x[8];
      const int k0 = 32 * c + 8 * q;
      const float* hp = a.h0 + (size_t)rowc * a.Ip + k0;
      if (k0 < a.Ip) {                                   // (Ip is a multiple of 16: the 8 values are all inside or all outside)
        const f4 u = *reinterpret_cast<const f4*>(hp), v = *reinterpret_cast<const f4*>(hp + 4);
        x[0] = u[0]; x[1] = u[1]; x[2] = u[2]; x[3] = u[3]; x[4] = v[0]; x[5] = v[1]; x[6] = v[2]; x[7] = v[3];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.f;
      }
      fx_split8(x, hb[c][0], hb[c][1], hb[c][2]);
    }
  }
  f4 w2v[NT1];
#pragma unroll
  for (int u = 0; u < NT1; ++u) w2v[u] = *reinterpret_cast<const f4*>(a.w2 + u * 16 + 4 * q);

  f4 ah[4], ac[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { ah[t] = f4{0, 0, 0, 0}; ac[t] = f4{0, 0, 0, 0}; }
  const int lofs = (q * 16 + i) * 16;                    // this lane's 16 bytes inside a block
  __syncthreads();
#pragma unroll
  for (int ph = 0; ph < FX_MAXC0; ++ph) {
    if (ph < nc0) {
      const unsigned char* cur = buf_of(ph);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        int tt = t0 + t; tt = tt < NT0 ? tt : NT0 - 1;   // (a wavefront with fewer tiles multiplies a throw-away one)
        fx_bf8 w[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) w[p] = *reinterpret_cast<const fx_bf8*>(cur + (size_t)(tt * 3 + p) * FX_BLOCK + lofs);
        fx_mma6(w, hb[ph], ah[t], ac[t]);
      }
      __syncthreads();
    }
  }
  // A0 = sigm(Z0); lane (i, q) holds Z0[row i][16 (t0 + t) + 4 q + r]
  float a0[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = (t0 + t) * 16 + 4 * q + r;
      const float sg = chain_sigm(ac[t][r] + ah[t][r]);
      a0[t][r] = (n < a.H1 && t < ntl) ? sg : 0.0f;
    }
  // layer 1: this wavefront's K range = its own tiles, two per chunk
  f4 bh[NT1], bc[NT1];
#pragma unroll
  for (int u = 0; u < NT1; ++u) { bh[u] = f4{0, 0, 0, 0}; bc[u] = f4{0, 0, 0, 0}; }
  int cc0 = 0;                                           // first chunk of this wavefront in the W1 image
  for (int w = 0; w < wave; ++w) { int nt = NT0 - 4 * w; nt = nt < 0 ? 0 : (nt > 4 ? 4 : nt); cc0 += (nt + 1) / 2; }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float x[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { x[r] = a0[2 * c][r]; x[4 + r] = a0[2 * c + 1][r]; }
    fx_bf8 xb[3];
    fx_split8(x, xb[0], xb[1], xb[2]);
    int cc = cc0 + c; cc = cc < ncc ? cc : ncc - 1;      // (past this wavefront's tiles the activations are all zero)
#pragma unroll
    for (int u = 0; u < NT1; ++u) {
      fx_bf8 w[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) w[p] = *reinterpret_cast<const fx_bf8*>(bufB + (size_t)((cc * NT1 + u) * 3 + p) * FX_BLOCK + lofs);
      fx_mma6(w, xb, bh[u], bc[u]);
    }
  }
  // partial Z1 of the four wavefronts: exchanged through LDS (bufA is free since the last W0 phase), added in wavefront order
  float* const xch = reinterpret_cast<float*>(bufA);
  constexpr int XS = NT1 * 4;
  {
    float* xw = xch + (wave * XS) * 64 + lane;
#pragma unroll
    for (int u = 0; u < NT1; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) xw[(u * 4 + r) * 64] = bc[u][r] + bh[u][r];
  }
  __syncthreads();
  float part = 0.f;
#pragma unroll
  for (int u = 0; u < NT1; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float z = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) z += xch[(w * XS + u * 4 + r) * 64 + lane];
      const int n = u * 16 + 4 * q + r;
      part += (n < a.H2 ? chain_sigm(z) : 0.0f) * w2v[u][r];
    }
  float z2 = part + __shfl_xor(part, 16, 64);
  z2 += __shfl_xor(z2, 32, 64);
  if (wave == 0 && q == 0 && vrow) a.yhat[row] = sigm_out(z2);
}

}  // namespace goctr
