// ctr_x3_images.h -- bf16-plane weight images of the 6-product-split chain kernel (ctr_chain_x3.h): layout index
// functions (host + device), the per-parameter scatter the Adam kernels call, and the rebuild-everything kernel.
//
// Every float32 weight w is stored as three bf16 planes, w = hi + mid + lo exactly (round-to-nearest splits), in the
// order the v_mfma_f32_32x32x16_bf16 A-operand fragments want them: one 16-byte entry = the 8 k-values of lane
// (m = lane % 32, kg = lane / 32) of one 32 x 16 block, 64 entries = 1 KiB = one coalesced wave load.
//   IMG0  W0   for F0:  [H1 tile t][k chunk c][plane][lane = 32 kg + m][8]   = W0[16c + 8kg + s][32t + m]
//   IMG1  W1   for F1:  [H1 chunk cc][H2 tile u][plane][lane = 32 kg + m][8] = W1[perm(cc, kg, s)][32u + m]
//   IMG2  W1^T for B0:  [H1 tile t][H2 chunk c][plane][lane = 32 kg + m][8]  = W1[32t + m][16c + 8kg + s]
//   IMG3  W0[U:U+D]^T for dp: [H1 chunk cc][plane][lane = 32 kg + d][8]      = W0[U + d][perm(cc, kg, s)]   (d < 2 D when 2 D <= 32)
// perm(cc, kg, s) = the H1 feature a lane holds at accumulator position of the 32x32 MFMA result:
//   32 (cc / 2) + 8 (2 (cc % 2) + s / 4) + 4 kg + s % 4.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace goctr {

constexpr int CX_NT0 = 7;     // 32-feature tiles of H1 (H1p <= 224)
constexpr int CX_NU = 3;      // 32-feature tiles of H2 (H2p <= 96)
constexpr int CX_NCH2 = 5;    // 16-k chunks of H2 (H2p == 80)
constexpr int CX_NCC = 14;    // 16-k chunks of H1 in accumulator order
constexpr int CX_PF = 3;      // chunks of A operands in flight

typedef __bf16 cx_bf8 __attribute__((ext_vector_type(8)));
typedef float cx_acc __attribute__((ext_vector_type(16)));
typedef float cx_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int cx_u4 __attribute__((ext_vector_type(4)));

typedef unsigned int cx_u2 __attribute__((ext_vector_type(2)));

// ---- image index functions (bf16 element units), shared by the builders and the tests of the layout
__host__ __device__ inline void cx_perm(int f1, int& cc, int& kg, int& s) {
  const int tt = f1 >> 5, g = (f1 >> 3) & 3, h = (f1 >> 2) & 1, r = f1 & 3;
  cc = 2 * tt + (g >> 1); kg = h; s = 4 * (g & 1) + r;
}
__host__ __device__ inline size_t cx_img0_index(int k, int f1, int nch0, int p) {
  return ((((size_t)(f1 >> 5) * nch0 + (k >> 4)) * 3 + p) * 64 + ((k >> 3) & 1) * 32 + (f1 & 31)) * 8 + (k & 7);
}
__host__ __device__ inline size_t cx_img1_index(int f1, int f2, int p) {
  int cc, kg, s; cx_perm(f1, cc, kg, s);
  return ((((size_t)cc * CX_NU + (f2 >> 5)) * 3 + p) * 64 + kg * 32 + (f2 & 31)) * 8 + s;
}
__host__ __device__ inline size_t cx_img2_index(int f1, int f2, int p) {
  return ((((size_t)(f1 >> 5) * CX_NCH2 + (f2 >> 4)) * 3 + p) * 64 + ((f2 >> 3) & 1) * 32 + (f1 & 31)) * 8 + (f2 & 7);
}
__host__ __device__ inline size_t cx_img3_index(int f1, int d, int p) {
  int cc, kg, s; cx_perm(f1, cc, kg, s);
  return (((size_t)cc * 3 + p) * 64 + kg * 32 + d) * 8 + s;
}
inline size_t cx_img0_elems(int nch0) { return (size_t)CX_NT0 * nch0 * 3 * 512; }
inline size_t cx_img1_elems() { return (size_t)CX_NCC * CX_NU * 3 * 512; }
inline size_t cx_img2_elems() { return (size_t)CX_NT0 * CX_NCH2 * 3 * 512; }
inline size_t cx_img3_elems() { return (size_t)CX_NCC * 3 * 512; }
inline size_t cx_images_elems(int nch0) { return cx_img0_elems(nch0) + cx_img1_elems() + cx_img2_elems() + cx_img3_elems(); }

// x = hi + mid + lo, each plane the round-to-nearest-even bf16 of what the previous ones left (exact for float32)
__device__ __forceinline__ void cx_split1(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
  const __bf16 h = (__bf16)x;
  const float r1 = x - (float)h;
  const __bf16 m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  const __bf16 l = (__bf16)r2;
  hi = __builtin_bit_cast(unsigned short, h); mid = __builtin_bit_cast(unsigned short, m); lo = __builtin_bit_cast(unsigned short, l);
}

// the bf16-plane fragment images of the 32-row chain kernels (training and predict)
struct CxImages {
  unsigned short* img0; unsigned short* img1; unsigned short* img2; unsigned short* img3; int nch0;
};

// the image entries of ONE parameter (called by the Adam kernels right after the update; idx = index in the padded flat
// weight buffer, see AdamArgs): three 2-byte stores per image the element appears in
__device__ __forceinline__ void cx_scatter_weight(const CxImages& im, float w, int idx, int off1, int off2, int H1p, int H2p,
                                                  int U, int D) {
  if (!im.img0) return;
  unsigned short pl[3];
  if (idx < off1) {
    const int k = idx / H1p, f1 = idx - k * H1p;
    cx_split1(w, pl[0], pl[1], pl[2]);
#pragma unroll
    for (int p = 0; p < 3; ++p) im.img0[cx_img0_index(k, f1, im.nch0, p)] = pl[p];
    // rows U .. U+D-1 give dp (d cost / d pooled); when 2 D <= 32 the 32-wide product has room for rows U+D .. U+2D-1 as well:
    // d cost / d candidate-item segment, which the trainable-embedding path needs (emb_train.h: dpv = [dp | dvh]) and used
    // to get from a GEMM launch of its own.  The frozen path stores only the first Dp columns.
    const int Dx = 2 * D <= 32 ? 2 * D : D;
    if (k >= U && k < U + Dx && D <= 32) {
#pragma unroll
      for (int p = 0; p < 3; ++p) im.img3[cx_img3_index(f1, k - U, p)] = pl[p];
    }
  } else if (idx < off2) {
    const int e = idx - off1;
    const int f1 = e / H2p, f2 = e - f1 * H2p;
    cx_split1(w, pl[0], pl[1], pl[2]);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      im.img1[cx_img1_index(f1, f2, p)] = pl[p];
      im.img2[cx_img2_index(f1, f2, p)] = pl[p];
    }
  }
}

// (re)build every image from the flat float32 weights: after a host upload of weights (the images start zeroed, and the
// padded weight entries are zero, so only real entries need writing -- but writing all keeps it simple)
#ifndef GOCTR_NO_PLAIN_KERNELS   // (a second translation unit includes this header for its templates only: ctr_fwd.hip)
__global__ __launch_bounds__(256) void x3_build_images_kernel(const float* __restrict__ W, int off1, int off2, int H1p, int H2p,
                                                              int U, int D, CxImages im) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= off2) return;
  cx_scatter_weight(im, W[idx], idx, off1, off2, H1p, H2p, U, D);
}
#endif


}  // namespace goctr
