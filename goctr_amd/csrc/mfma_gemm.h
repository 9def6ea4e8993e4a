// mfma_gemm.h -- skinny-GEMM device templates for gfx950 on the exact-precision MFMA forms:
//   float  : v_mfma_f32_16x16x4_f32  (exact f32 fmaf chain, 157 TF chip peak)
//   double : v_mfma_f64_16x16x4_f64  (sklearn-port MLP, which is float64 in the reference)
//
// Shapes on go-ctr's path are "tall and skinny": M = batch (4k-16k rows), K,N <= a few hundred
// (model/din/din.go:17-18: 137->200->80->1; nn MLP 281->100->1).  Two kernels cover them:
//
//   gemm_nn : C[M,N] = epi(A[M,K] . B[K,N])   forward layers and backward-data (B = W^T copy)
//             one workgroup = 4 wavefronts = (4/WN) row strips of 16 rows x WN column groups;
//             B is streamed through LDS in 16-row K-chunks (double buffered), the A fragment is
//             one 16-byte global load per lane per chunk (k-permuted so the 4 values feed 4 MFMAs).
//   gemm_tn : dW[K,N] = sum_m A[m,K]^T . D[m,N] split over M (weight gradients); every workgroup
//             owns a contiguous row range, stages 16-row chunks of A and D in LDS and writes one
//             partial slab; slabs are summed in a fixed order by the reduce kernel (deterministic).
//
// Leading dimensions are multiples of 16 elements and all pad entries are zero, so there is no
// bounds handling on K or N inside the loops.
#pragma once
#include <hip/hip_runtime.h>

namespace goctr {

template <typename T> struct Mfma;

template <> struct Mfma<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  typedef float vec_t __attribute__((ext_vector_type(4)));  // 16-byte vector
  static constexpr int VEC = 4;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout: col = lane&15, row = 4*(lane>>4) + r
  static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
  static __device__ __forceinline__ void load4(const float* p, float out[4]) {
    vec_t v = *reinterpret_cast<const vec_t*>(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  }
};

template <> struct Mfma<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  typedef double vec_t __attribute__((ext_vector_type(2)));  // 16-byte vector
  static constexpr int VEC = 2;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // f64 C/D layout differs from f32: col = lane&15, row = (lane>>4) + 4*r
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
  static __device__ __forceinline__ void load4(const double* p, double out[4]) {
    vec_t v0 = *reinterpret_cast<const vec_t*>(p);
    vec_t v1 = *reinterpret_cast<const vec_t*>(p + 2);
    out[0] = v0.x; out[1] = v0.y; out[2] = v1.x; out[3] = v1.y;
  }
};

// LDS row stride (in elements) for the TN kernel's [16][cols] tiles: the two 16-lane groups of a
// half wave read rows m and m+1, so the stride must be == 16 (mod 32) elements to land the second
// row on the other half of the banks (ds_read_b32: 32 banks; ds_read_b64: 64 dword banks).
__host__ __device__ inline int tn_lds_stride(int cols) { return cols + ((16 - cols % 32) + 32) % 32; }

constexpr int GEMM_NN_NTW = 7;     // n-tiles (16 cols each) per wavefront
constexpr int GEMM_NN_MAXV = 8;    // max 16-byte vectors a thread stages per K-chunk

template <typename T>
inline size_t gemm_nn_lds_bytes(int ncols_blk) { return (size_t)2 * 16 * (ncols_blk + 4) * sizeof(T); }

// C = epi(A . B).  grid = (ceil(M / (16*WM)), ceil(NT / (WN*NTW))), block = 256, WM = 4 / WN.
// epi(row, col, value) is called once per output element of rows < M.
template <typename T, class Epi>
__global__ __launch_bounds__(256, 4) void gemm_nn_kernel(const T* __restrict__ A, int lda,
                                                      const T* __restrict__ Bm, int ldb, int M, int Kp,
                                                      int Np, int WN, Epi epi) {
  using MF = Mfma<T>;
  using acc_t = typename MF::acc_t;
  using vec_t = typename MF::vec_t;
  constexpr int VEC = MF::VEC;
  constexpr int NTW = GEMM_NN_NTW;
  extern __shared__ __attribute__((aligned(16))) unsigned char goctr_smem[];
  T* Bs = reinterpret_cast<T*>(goctr_smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int WM = 4 / WN;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int i = lane & 15, q = lane >> 4;
  const int NTall = Np >> 4;
  const int nblk0 = blockIdx.y * (WN * NTW);
  int nt_blk = NTall - nblk0;
  if (nt_blk > WN * NTW) nt_blk = WN * NTW;
  const int ncols_blk = nt_blk * 16;
  const int Ns = ncols_blk + 4;  // Ns % 8 == 4: rows k and k+4 sit on opposite bank halves
  int ntiles = nt_blk - wn * NTW;
  ntiles = ntiles < 0 ? 0 : (ntiles > NTW ? NTW : ntiles);
  const int row0 = blockIdx.x * (16 * WM) + wm * 16;
  int arow = row0 + i;
  if (arow > M - 1) arow = M - 1;
  const T* ap = A + (size_t)arow * lda + 4 * q;

  acc_t acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = acc_t{0, 0, 0, 0};

  const int nchunks = Kp >> 4;
  const int vpr = ncols_blk / VEC;  // 16-byte vectors per staged row
  const int total_v = 16 * vpr;
  int goff[GEMM_NN_MAXV], loff[GEMM_NN_MAXV];
#pragma unroll
  for (int s = 0; s < GEMM_NN_MAXV; ++s) {
    int idx = tid + s * 256;
    int r = idx / vpr, cv = idx - r * vpr;
    goff[s] = r * ldb + nblk0 * 16 + cv * VEC;
    loff[s] = r * Ns + cv * VEC;
  }
  auto stage = [&](int c, int buf) {
    const T* src = Bm + (size_t)c * 16 * ldb;
    T* dst = Bs + (size_t)buf * 16 * Ns;
#pragma unroll
    for (int s = 0; s < GEMM_NN_MAXV; ++s)
      if (tid + s * 256 < total_v)
        *reinterpret_cast<vec_t*>(dst + loff[s]) = *reinterpret_cast<const vec_t*>(src + goff[s]);
  };

  stage(0, 0);
  __syncthreads();
  int buf = 0;
  for (int c = 0; c < nchunks; ++c) {
    T av[4];
    MF::load4(ap + c * 16, av);
    if (c + 1 < nchunks) stage(c + 1, buf ^ 1);
    const T* bsb = Bs + (size_t)buf * 16 * Ns + wn * NTW * 16 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const T* brow = bsb + (4 * q + j) * Ns;
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        if (t < ntiles) acc[t] = MF::mma(av[j], brow[t * 16], acc[t]);
    }
    __syncthreads();
    buf ^= 1;
  }

  const int nt0 = nblk0 + wn * NTW;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    if (t < ntiles) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + MF::crow(lane, r);
        if (row < M) epi(row, (nt0 + t) * 16 + i, acc[t][r]);
      }
    }
  }
}

// Weight-gradient GEMM: slab[blockIdx.x][k][n] = sum over this workgroup's rows of A[m][k]*D[m][n].
// block = 64*WK*WN threads; grid = (M-splits, k-blocks, n-blocks); a k-block is WK*KTW tiles of 16
// columns of A, an n-block WN*NTW tiles of D; wave (wk,wn) owns KTW x NTW output tiles of it.
// KT / NT = total number of 16-wide tiles of A's / D's columns; the slab is [KT*16][NT*16].
template <typename T, int KTW, int NTW>
__global__ __launch_bounds__(1024) void gemm_tn_kernel(const T* __restrict__ A, int lda, int KT,
                                                       const T* __restrict__ Dm, int ldd, int NT, int M,
                                                       int rows_per_wg, int WK, int WN,
                                                       T* __restrict__ slabs, size_t slab_stride) {
  using MF = Mfma<T>;
  using acc_t = typename MF::acc_t;
  using vec_t = typename MF::vec_t;
  constexpr int VEC = MF::VEC;
  extern __shared__ __attribute__((aligned(16))) unsigned char goctr_smem[];
  const int kb0 = blockIdx.y * WK * KTW, nb0 = blockIdx.z * WN * NTW;
  int kb_t = KT - kb0; if (kb_t > WK * KTW) kb_t = WK * KTW;
  int nb_t = NT - nb0; if (nb_t > WN * NTW) nb_t = WN * NTW;
  const int Kc = kb_t * 16, Nc = nb_t * 16;
  const int Kas = tn_lds_stride(Kc), Nds = tn_lds_stride(Nc);
  T* As = reinterpret_cast<T*>(goctr_smem);                 // [2][16][Kas]
  T* Ds = As + 2 * 16 * tn_lds_stride(WK * KTW * 16);       // [2][16][Nds]

  const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / WN, wn = wave - wk * WN;
  const int i = lane & 15, q = lane >> 4;
  const int kt0 = wk * KTW, nt0 = wn * NTW;  // tile offsets inside the block
  int kcnt = kb_t - kt0; kcnt = kcnt < 0 ? 0 : (kcnt > KTW ? KTW : kcnt);
  int ncnt = nb_t - nt0; ncnt = ncnt < 0 ? 0 : (ncnt > NTW ? NTW : ncnt);
  const int m_begin = blockIdx.x * rows_per_wg;
  int m_end = m_begin + rows_per_wg;
  if (m_end > M) m_end = M;

  acc_t acc[KTW][NTW];
#pragma unroll
  for (int e = 0; e < KTW; ++e)
#pragma unroll
    for (int f = 0; f < NTW; ++f) acc[e][f] = acc_t{0, 0, 0, 0};

  const int kv = Kc / VEC, nv = Nc / VEC;
  const T* Ab = A + kb0 * 16;
  const T* Db = Dm + nb0 * 16;
  auto stage = [&](int m0, int buf) {
    T* as = As + (size_t)buf * 16 * Kas;
    T* ds = Ds + (size_t)buf * 16 * Nds;
    for (int idx = tid; idx < 16 * kv; idx += nthreads) {
      int r = idx / kv, cv = idx - r * kv;
      vec_t v = vec_t(0);
      if (m0 + r < m_end) v = *reinterpret_cast<const vec_t*>(Ab + (size_t)(m0 + r) * lda + cv * VEC);
      *reinterpret_cast<vec_t*>(as + r * Kas + cv * VEC) = v;
    }
    for (int idx = tid; idx < 16 * nv; idx += nthreads) {
      int r = idx / nv, cv = idx - r * nv;
      vec_t v = vec_t(0);
      if (m0 + r < m_end) v = *reinterpret_cast<const vec_t*>(Db + (size_t)(m0 + r) * ldd + cv * VEC);
      *reinterpret_cast<vec_t*>(ds + r * Nds + cv * VEC) = v;
    }
  };

  if (m_begin < m_end) {
    stage(m_begin, 0);
    __syncthreads();
    int buf = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += 16) {
      if (m0 + 16 < m_end) stage(m0 + 16, buf ^ 1);
      const T* as = As + (size_t)buf * 16 * Kas + kt0 * 16 + i;
      const T* ds = Ds + (size_t)buf * 16 * Nds + nt0 * 16 + i;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        T a[KTW], d[NTW];
#pragma unroll
        for (int e = 0; e < KTW; ++e) a[e] = e < kcnt ? as[(4 * s + q) * Kas + e * 16] : T(0);
#pragma unroll
        for (int f = 0; f < NTW; ++f) d[f] = f < ncnt ? ds[(4 * s + q) * Nds + f * 16] : T(0);
#pragma unroll
        for (int e = 0; e < KTW; ++e)
#pragma unroll
          for (int f = 0; f < NTW; ++f)
            if (e < kcnt && f < ncnt) acc[e][f] = MF::mma(a[e], d[f], acc[e][f]);
      }
      __syncthreads();
      buf ^= 1;
    }
  }

  T* out = slabs + (size_t)blockIdx.x * slab_stride;
  const int ldo = NT * 16;
#pragma unroll
  for (int e = 0; e < KTW; ++e)
#pragma unroll
    for (int f = 0; f < NTW; ++f)
      if (e < kcnt && f < ncnt) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[(size_t)((kb0 + kt0 + e) * 16 + MF::crow(lane, r)) * ldo + (nb0 + nt0 + f) * 16 + i] = acc[e][f][r];
      }
}

// LDS bytes for a (WK*KTW) x (WN*NTW)-tile block
template <typename T>
inline size_t gemm_tn_lds_bytes(int kb_tiles, int nb_tiles) {
  return (size_t)2 * 16 * (tn_lds_stride(kb_tiles * 16) + tn_lds_stride(nb_tiles * 16)) * sizeof(T);
}

}  // namespace goctr
