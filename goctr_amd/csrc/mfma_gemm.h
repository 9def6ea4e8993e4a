// mfma_gemm.h -- skinny-GEMM device templates for gfx950 on the exact-precision MFMA forms:
//   float  : v_mfma_f32_16x16x4_f32  (exact f32 fmaf chain, 157 TF chip peak)
//   double : v_mfma_f64_16x16x4_f64  (sklearn-port MLP, which is float64 in the reference)
//
// Shapes on go-ctr's path are "tall and skinny": M = batch (4k-16k rows), K,N <= a few hundred
// (model/din/din.go:17-18: 137->200->80->1; nn MLP 281->100->1).  Two kernels cover them:
//
//   gemm_nn : C[M,N] = epi(A[M,K] . B[K,N])   forward layers and backward-data (B = W^T copy).
//             B is small (<= 120 KB): a workgroup (4 wavefronts) parks the WHOLE of B (or a K-phase
//             of it) in LDS with one burst of 16-byte loads, fetches all of its A fragments with
//             one burst of 16-byte global loads per lane (k-permuted so the 4 values of a load
//             feed 4 consecutive MFMAs) and then runs an LDS->MFMA loop with no global traffic and
//             no barrier.  One barrier per phase instead of one per 16-deep K chunk.
//   gemm_tn : dW[K,N] = sum_m A[m,K]^T . D[m,N] split over M (weight gradients); a workgroup owns a
//             contiguous row range and a block of the output, streams CH-row chunks of A and D
//             through registers into double-buffered LDS (global loads of chunk c+1 are in flight
//             while chunk c is multiplied) and writes one partial slab; slabs are summed in a fixed
//             order by the reduce kernel (deterministic).
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

// LDS row stride (in elements) for the TN kernel's [CH][cols] tiles: the two 16-lane groups of a
// half wave read rows m and m+1, so the stride must be == 16 (mod 32) elements to land the second
// row on the other half of the banks (ds_read_b32: 32 banks; ds_read_b64: 64 dword banks).
__host__ __device__ inline int tn_lds_stride(int cols) { return cols + ((16 - cols % 32) + 32) % 32; }

constexpr int GEMM_NN_NTW = 7;     // max n-tiles (16 cols each) per wavefront
// max 16-deep K chunks per phase (the A fragments of a phase live in registers: 4 values per chunk)
template <typename T> struct NnPhase { static constexpr int CH = sizeof(T) == 4 ? 16 : 8; };
constexpr size_t GEMM_LDS_BUDGET = 152 * 1024;  // of the CU's 160 KiB

// rows of B one phase can park in LDS for a block of ncols_blk columns
template <typename T>
inline int gemm_nn_phase_rows(int Kp, int ncols_blk) {
  int rows = (int)(GEMM_LDS_BUDGET / ((size_t)(ncols_blk + 4) * sizeof(T))) / 16 * 16;
  if (rows > NnPhase<T>::CH * 16) rows = NnPhase<T>::CH * 16;
  return rows < Kp ? rows : Kp;
}
template <typename T>
inline size_t gemm_nn_lds_bytes(int kph, int ncols_alloc) { return (size_t)kph * (ncols_alloc + 4) * sizeof(T); }

// The same product for a B operand that fits one LDS phase (Kp <= KPH), PERSISTENT over row tiles: a workgroup parks B once
// and then walks row tiles blockIdx.x, + gridDim.x, ...; the A fragments of the NEXT tile are requested before the current
// tile's MFMAs (one workgroup per CU -- B takes most of the LDS -- so no other wavefront would hide that latency).
// gemm_nn_kernel parks B once per 16*WM rows: at M = 16384, Kp = 208, Np = 128 (the dpv product of the trainable-embedding
// path at cfg4) that was 512 fills of 106 KB, each in front of 208 MFMAs per wavefront: 29 us, MFMA-busy 19 %.
template <typename T, class Epi, int NTW>
__global__ __launch_bounds__(256, 1) void gemm_nn_rows_kernel(const T* __restrict__ A, int lda, const T* __restrict__ Bm, int ldb, int M,
                                                              int Kp, int Np, int WN, Epi epi) {
  constexpr int ntw = NTW;
  using MF = Mfma<T>;
  using acc_t = typename MF::acc_t;
  using vec_t = typename MF::vec_t;
  constexpr int VEC = MF::VEC;
  constexpr int PHCH = NnPhase<T>::CH;
  extern __shared__ __attribute__((aligned(16))) unsigned char goctr_smem[];
  T* Bs = reinterpret_cast<T*>(goctr_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int WM = 4 / WN;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int i = lane & 15, q = lane >> 4;
  const int NTall = Np >> 4;
  const int nblk0 = blockIdx.y * (WN * ntw);
  int nt_blk = NTall - nblk0;
  if (nt_blk > WN * ntw) nt_blk = WN * ntw;
  const int ncols_blk = nt_blk * 16;
  const int Ns = WN * ntw * 16 + 4;
  int ntiles = nt_blk - wn * ntw;
  ntiles = ntiles < 0 ? 0 : (ntiles > ntw ? ntw : ntiles);
  const int nch = Kp >> 4;
  // park B once
  {
    const int vpr = ncols_blk / VEC, total_v = Kp * vpr;
    const T* bsrc = Bm + nblk0 * 16;
    for (int idx = tid; idx < total_v; idx += 256) {
      const int r = idx / vpr, cv = idx - r * vpr;
      *reinterpret_cast<vec_t*>(Bs + r * Ns + cv * VEC) = *reinterpret_cast<const vec_t*>(bsrc + (size_t)r * ldb + cv * VEC);
    }
  }
  const int nrt = (M + 16 * WM - 1) / (16 * WM);
  auto load_a = [&](int rt, T (&dst)[PHCH][4]) {
    int arow = rt * (16 * WM) + wm * 16 + i;
    if (arow > M - 1) arow = M - 1;
    const T* ap = A + (size_t)arow * lda + 4 * q;
#pragma unroll
    for (int c = 0; c < PHCH; ++c)
      if (c < nch) MF::load4(ap + c * 16, dst[c]);
  };
  T av[PHCH][4], an[PHCH][4];
  int rt = blockIdx.x;
  if (rt < nrt) load_a(rt, av);
  __syncthreads();
  for (; rt < nrt; rt += gridDim.x) {
    const int nxt = rt + gridDim.x;
    if (nxt < nrt) load_a(nxt, an);
    acc_t acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = acc_t{0, 0, 0, 0};
    const T* brow = Bs + (4 * q) * Ns + wn * ntw * 16 + i;
#pragma unroll
    for (int c = 0; c < PHCH; ++c) {
      if (c < nch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[t] = MF::mma(av[c][j], brow[t * 16], acc[t]);
          brow += Ns;
        }
        brow += 12 * Ns;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int row0 = rt * (16 * WM) + wm * 16, nt0 = nblk0 + wn * ntw;
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
#pragma unroll
    for (int c = 0; c < PHCH; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) av[c][j] = an[c][j];
  }
}

// C = epi(A . B).  grid = (ceil(M / (16*WM)), n-blocks), block = 256, WM = 4 / WN; a column block is
// WN*ntw tiles, wave (wm,wn) owns rows [16*wm,+16) x tiles [wn*ntw, +ntw).
// epi(row, col, value) is called once per output element of rows < M.
template <typename T, class Epi, int NTW>
__global__ __launch_bounds__(256, 2) void gemm_nn_kernel(const T* __restrict__ A, int lda,
                                                         const T* __restrict__ Bm, int ldb, int M, int Kp,
                                                         int Np, int WN, int KPH, Epi epi) {
  constexpr int ntw = NTW;
  using MF = Mfma<T>;
  using acc_t = typename MF::acc_t;
  using vec_t = typename MF::vec_t;
  constexpr int VEC = MF::VEC;
  constexpr int GEMM_NN_PHCH = NnPhase<T>::CH;
  extern __shared__ __attribute__((aligned(16))) unsigned char goctr_smem[];
  T* Bs = reinterpret_cast<T*>(goctr_smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int WM = 4 / WN;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int i = lane & 15, q = lane >> 4;
  const int NTall = Np >> 4;
  const int nblk0 = blockIdx.y * (WN * ntw);
  int nt_blk = NTall - nblk0;
  if (nt_blk > WN * ntw) nt_blk = WN * ntw;
  const int ncols_blk = nt_blk * 16;
  // LDS rows are WN*NTW tiles wide even when the block has fewer: edge waves then read (and discard)
  // unfilled columns instead of needing per-tile address clamps.  Ns % 8 == 4: rows k and k+4 sit
  // on opposite bank halves.
  const int Ns = WN * ntw * 16 + 4;
  int ntiles = nt_blk - wn * ntw;
  ntiles = ntiles < 0 ? 0 : (ntiles > ntw ? ntw : ntiles);
  const int row0 = blockIdx.x * (16 * WM) + wm * 16;
  int arow = row0 + i;
  if (arow > M - 1) arow = M - 1;
  const T* ap = A + (size_t)arow * lda + 4 * q;

  acc_t acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = acc_t{0, 0, 0, 0};

  const int vpr = ncols_blk / VEC;  // 16-byte vectors per staged row
  for (int k0 = 0; k0 < Kp; k0 += KPH) {
    const int kph = Kp - k0 < KPH ? Kp - k0 : KPH;
    const int nch = kph >> 4;
    // all A fragments of this phase: one burst of 16-byte loads per lane
    T av[GEMM_NN_PHCH][4];
#pragma unroll
    for (int c = 0; c < GEMM_NN_PHCH; ++c)
      if (c < nch) MF::load4(ap + k0 + c * 16, av[c]);
    if (k0 > 0) __syncthreads();  // the previous phase's LDS reads are done
    // park B[k0:k0+kph, block columns] in LDS
    const int total_v = kph * vpr;
    const T* bsrc = Bm + (size_t)k0 * ldb + nblk0 * 16;
    for (int base = 0; base < total_v; base += 256 * 8) {
      vec_t tmp[8];
      int lo[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int idx = base + tid + s * 256;
        if (idx < total_v) {
          const int r = idx / vpr, cv = idx - r * vpr;
          tmp[s] = *reinterpret_cast<const vec_t*>(bsrc + (size_t)r * ldb + cv * VEC);
          lo[s] = r * Ns + cv * VEC;
        }
      }
#pragma unroll
      for (int s = 0; s < 8; ++s)
        if (base + tid + s * 256 < total_v) *reinterpret_cast<vec_t*>(Bs + lo[s]) = tmp[s];
    }
    __syncthreads();
    // one row pointer walks down the staged rows; the tile offsets are ds_read immediates
    const T* brow = Bs + (4 * q) * Ns + wn * ntw * 16 + i;
#pragma unroll
    for (int c = 0; c < GEMM_NN_PHCH; ++c) {
      if (c < nch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int t = 0; t < NTW; ++t) acc[t] = MF::mma(av[c][j], brow[t * 16], acc[t]);
          brow += Ns;
        }
        brow += 12 * Ns;
        // keep the scheduler from hoisting every later chunk's ds_reads above this chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  const int nt0 = nblk0 + wn * ntw;
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
// CH = rows per pipelined chunk (multiple of 16).
constexpr int GEMM_TN_MAXVA = 8, GEMM_TN_MAXVD = 8;

template <typename T, int KTW, int NTW, int CH>
__global__ __launch_bounds__(512) void gemm_tn_kernel(const T* __restrict__ A, int lda, int KT,
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
  // strides cover the full WK*KTW / WN*NTW tile block so edge waves read in-bounds (unused) data
  const int Kas = tn_lds_stride(WK * KTW * 16), Nds = tn_lds_stride(WN * NTW * 16);
  T* As = reinterpret_cast<T*>(goctr_smem);                 // [2][CH][Kas]
  T* Ds = As + 2 * CH * Kas;                                // [2][CH][Nds]

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
  vec_t ra[GEMM_TN_MAXVA], rd[GEMM_TN_MAXVD];
  // global -> registers (issued early, consumed late)
  auto gload = [&](int m0) {
#pragma unroll
    for (int s = 0; s < GEMM_TN_MAXVA; ++s) {
      const int idx = tid + s * nthreads;
      ra[s] = vec_t(0);
      if (idx < CH * kv) {
        const int r = idx / kv, cv = idx - r * kv;
        if (m0 + r < m_end) ra[s] = *reinterpret_cast<const vec_t*>(Ab + (size_t)(m0 + r) * lda + cv * VEC);
      }
    }
#pragma unroll
    for (int s = 0; s < GEMM_TN_MAXVD; ++s) {
      const int idx = tid + s * nthreads;
      rd[s] = vec_t(0);
      if (idx < CH * nv) {
        const int r = idx / nv, cv = idx - r * nv;
        if (m0 + r < m_end) rd[s] = *reinterpret_cast<const vec_t*>(Db + (size_t)(m0 + r) * ldd + cv * VEC);
      }
    }
  };
  auto lstore = [&](int buf) {
    T* as = As + (size_t)buf * CH * Kas;
    T* ds = Ds + (size_t)buf * CH * Nds;
#pragma unroll
    for (int s = 0; s < GEMM_TN_MAXVA; ++s) {
      const int idx = tid + s * nthreads;
      if (idx < CH * kv) {
        const int r = idx / kv, cv = idx - r * kv;
        *reinterpret_cast<vec_t*>(as + r * Kas + cv * VEC) = ra[s];
      }
    }
#pragma unroll
    for (int s = 0; s < GEMM_TN_MAXVD; ++s) {
      const int idx = tid + s * nthreads;
      if (idx < CH * nv) {
        const int r = idx / nv, cv = idx - r * nv;
        *reinterpret_cast<vec_t*>(ds + r * Nds + cv * VEC) = rd[s];
      }
    }
  };

  if (m_begin < m_end) {
    gload(m_begin);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += CH) {
      const bool more = m0 + CH < m_end;
      if (more) gload(m0 + CH);
      const T* as = As + (size_t)buf * CH * Kas + q * Kas + kt0 * 16 + i;
      const T* ds = Ds + (size_t)buf * CH * Nds + q * Nds + nt0 * 16 + i;
#pragma unroll
      for (int s = 0; s < CH / 4; ++s) {
        T a[KTW], d[NTW];
#pragma unroll
        for (int e = 0; e < KTW; ++e) a[e] = as[e * 16];   // tiles past the block edge hold stale/zero
#pragma unroll
        for (int f = 0; f < NTW; ++f) d[f] = ds[f * 16];   // data: their products are never stored
#pragma unroll
        for (int e = 0; e < KTW; ++e)
#pragma unroll
          for (int f = 0; f < NTW; ++f) acc[e][f] = MF::mma(a[e], d[f], acc[e][f]);
        as += 4 * Kas;
        ds += 4 * Nds;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) lstore(buf ^ 1);
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

// LDS bytes for a (WK*KTW) x (WN*NTW)-tile block with CH-row chunks
template <typename T>
inline size_t gemm_tn_lds_bytes(int kb_tiles, int nb_tiles, int ch) {
  return (size_t)2 * ch * (tn_lds_stride(kb_tiles * 16) + tn_lds_stride(nb_tiles * 16)) * sizeof(T);
}

// ---------------------------------------------------------------------------------------------------
// Several weight-gradient GEMMs in ONE launch (they are independent, individually too small to fill the
// chip, and each pays the same load -> LDS -> MFMA -> slab-store latency chain: run them side by side).
// Every workgroup = 4 wavefronts = 1 x 4 waves of KTW x NTW tiles: a k-block of KTW tiles of A's columns
// against up to 4*NTW tiles of D's columns, over `rows_per_wg` batch rows.  A problem may be posed
// transposed (A and D swapped) so that its tile grid suits this shape; `transpose_out` then stores
// out[k][n] at slab[n*ld_out + k] (one 16-byte store per lane: the f32 C layout has 4 consecutive k).
struct TnProblem {
  const float* A; int lda; int KT;
  const float* D; int ldd; int NT;
  float* slabs; unsigned long long slab_stride;
  int transpose_out; int ld_out;
  int first;   // first blockIdx.x of this problem; it owns kblocks*S blocks
  int rows;    // batch rows per workgroup (slab height) of this problem
  int S;       // slabs = ceil(M / rows)
  int nnb;     // (x3 wide blocks) n-blocks per k-block: the D tiles are split into nnb groups of nbt tiles; 0 = one group
  int nbt;
};
// wt (bf16-split bodies): slab stores go THROUGH the L2 (global_store ... sc1) instead of staying dirty in it until the launch
// ends -- 1: the 16-byte stores of the transposed problems, 2: also the 4-byte stores of the untransposed ones
struct TnMulti { TnProblem p[4]; int np; int M; unsigned long long* dbg; int wt; };

// The multi-problem weight-gradient kernel.  Per workgroup: one problem, one block of KTW (3, or 1 for the
// one-tile problems) 16-column tiles of A, ALL 16-column tiles of D split NTW = 4 per wavefront, one slab of
// rows_per_wg batch rows.  The reduction index of  slab[k][n] = sum_m A[m][k] * D[m][n]  is the ROW of both
// operands, so the staging transposes: a thread loads a 4-row x 4-column block (4 coalesced 16-byte loads),
// and writes it as 4 ds_write_b128 into column-major LDS strips T[col][m] (row stride CH + 4 floats = 16 B
// mod 128 B: the 16 lanes of a q-group read distinct bank groups).  One ds_read_b128 then holds the 4 rows a
// lane feeds to 4 consecutive MFMAs: 7 LDS instructions per 48 MFMAs instead of 7 per 12 (on gfx950 every
// LDS return stalls the fp32 MFMA stream of its SIMD, scripts/ubench/mfma_dma_overlap.hip).
template <int KTW, int NTW, int CH>
__device__ __forceinline__ void tn_multi_body(const TnMulti& a, const TnProblem& P, int kb, int split, unsigned char* smem) {
  using MF = Mfma<float>;
  typedef float acc_t __attribute__((ext_vector_type(4)));
  typedef float vec_t __attribute__((ext_vector_type(4)));
  constexpr int CHS = CH + 4;            // LDS floats per strip column
  constexpr int MAXB = ((CH / 4) * (KTW * 4 + 16 * 4) + 255) / 256;   // 4x4 staging blocks per thread and chunk (NT <= 16)
  const int kb0 = kb * KTW;
  int kb_t = P.KT - kb0; if (kb_t > KTW) kb_t = KTW;
  const int nb_t = P.NT;                 // <= 4*NTW (host-checked)
  const int Kc = kb_t * 16, Nc = nb_t * 16;
  const int kv = Kc >> 2, nv = Nc >> 2;
  float* As = reinterpret_cast<float*>(smem);       // [2][KTW*16][CHS]
  float* Ds = As + 2 * KTW * 16 * CHS;              // [2][Nc][CHS]
  const int a_buf = KTW * 16 * CHS, d_buf = Nc * CHS;

  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int nt0 = wn * NTW;
  int ncnt = nb_t - nt0; ncnt = ncnt < 0 ? 0 : (ncnt > NTW ? NTW : ncnt);
  const int m_begin = split * P.rows;
  int m_end = m_begin + P.rows;
  if (m_end > a.M) m_end = a.M;

  acc_t acc[KTW][NTW];
#pragma unroll
  for (int e = 0; e < KTW; ++e)
#pragma unroll
    for (int f = 0; f < NTW; ++f) acc[e][f] = acc_t{0, 0, 0, 0};

  // staging slots of this thread: block b = tid + 256 s  ->  (strip, row group rg, column group cg)
  const int nA = (CH / 4) * kv, nAll = nA + (CH / 4) * nv;
  const float* gsrc[MAXB]; int ld[MAXB]; int rg[MAXB]; int lofs[MAXB];   // lofs < 0: slot unused
  {
    const float rkv = 1.0f / (float)kv, rnv = 1.0f / (float)nv;
#pragma unroll
    for (int s = 0; s < MAXB; ++s) {
      const int b = tid + s * 256;
      gsrc[s] = P.A; ld[s] = P.lda; rg[s] = 0; lofs[s] = -1;
      if (b < nA) {
        const int r = (int)(((float)b + 0.5f) * rkv), cg = b - r * kv;       // exact for b < 2^20
        rg[s] = r; ld[s] = P.lda;
        gsrc[s] = P.A + kb0 * 16 + cg * 4;
        lofs[s] = (cg * 4) * CHS + 4 * r;
      } else if (b < nAll) {
        const int bb = b - nA;
        const int r = (int)(((float)bb + 0.5f) * rnv), cg = bb - r * nv;
        rg[s] = r; ld[s] = P.ldd;
        gsrc[s] = P.D + cg * 4;
        lofs[s] = 2 * a_buf + (cg * 4) * CHS + 4 * r;                         // relative to As
      }
    }
  }
  // Loads are unconditional (rows past the slab's end re-read its last row and are zeroed when the block is
  // written to LDS): no predicate, no register merge, hence no s_waitcnt between the loads of a chunk.
  vec_t st[MAXB][4];
  auto gload = [&](int m0) {
#pragma unroll
    for (int s = 0; s < MAXB; ++s) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int rr = m0 + 4 * rg[s] + r;
        rr = rr < m_end ? rr : m_end - 1;
        st[s][r] = *reinterpret_cast<const vec_t*>(gsrc[s] + (size_t)rr * ld[s]);
      }
    }
  };
  auto lstore = [&](int buf, int m0) {
#pragma unroll
    for (int s = 0; s < MAXB; ++s) {
      if (lofs[s] >= 0) {
        float* d = As + lofs[s] + (lofs[s] >= 2 * a_buf ? buf * d_buf : buf * a_buf);
        if (m0 + CH <= m_end) {                      // whole chunk inside the slab (wave-uniform)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            *reinterpret_cast<vec_t*>(d + c * CHS) = vec_t{st[s][0][c], st[s][1][c], st[s][2][c], st[s][3][c]};
        } else {
          const int left = m_end - (m0 + 4 * rg[s]);   // rows of this block that exist
#pragma unroll
          for (int c = 0; c < 4; ++c)
            *reinterpret_cast<vec_t*>(d + c * CHS) = vec_t{left > 0 ? st[s][0][c] : 0.f, left > 1 ? st[s][1][c] : 0.f,
                                                           left > 2 ? st[s][2][c] : 0.f, left > 3 ? st[s][3][c] : 0.f};
        }
      }
    }
  };
  // per-lane LDS offsets of the operand columns (tiles past the problem's edge re-read the last column)
  int aofs[KTW], dofs[NTW];
#pragma unroll
  for (int e = 0; e < KTW; ++e) { int c = e * 16 + i; c = c < Kc ? c : Kc - 1; aofs[e] = c * CHS + 4 * q; }
#pragma unroll
  for (int f = 0; f < NTW; ++f) { int c = (nt0 + f) * 16 + i; c = c < Nc ? c : Nc - 1; dofs[f] = c * CHS + 4 * q; }

  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, tsm = 0;
  if (a.dbg) ts0 = __builtin_amdgcn_s_memtime();
  if (m_begin < m_end) {
    gload(m_begin);
    lstore(0, m_begin);
    __syncthreads();
    if (a.dbg) ts1 = __builtin_amdgcn_s_memtime();
    int buf = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += CH) {
      const bool more = m0 + CH < m_end;
      if (more) gload(m0 + CH);
      const float* as = As + buf * a_buf;
      const float* ds = Ds + buf * d_buf;
#pragma unroll
      for (int g = 0; g < CH / 16; ++g) {
        vec_t av[KTW], dv[NTW];
#pragma unroll
        for (int e = 0; e < KTW; ++e) av[e] = *reinterpret_cast<const vec_t*>(as + aofs[e] + g * 16);
#pragma unroll
        for (int f = 0; f < NTW; ++f) dv[f] = *reinterpret_cast<const vec_t*>(ds + dofs[f] + g * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int e = 0; e < KTW; ++e)
#pragma unroll
            for (int f = 0; f < NTW; ++f) acc[e][f] = MF::mma(av[e][r], dv[f][r], acc[e][f]);
      }
      if (a.dbg && m0 == m_begin) tsm = __builtin_amdgcn_s_memtime();
      if (more) lstore(buf ^ 1, m0 + CH);
      __syncthreads();
      buf ^= 1;
    }
  }

  if (a.dbg) ts2 = __builtin_amdgcn_s_memtime();
  float* out = P.slabs + (size_t)split * P.slab_stride;
#pragma unroll
  for (int e = 0; e < KTW; ++e) {
#pragma unroll
    for (int f = 0; f < NTW; ++f) {
      if (e < kb_t && f < ncnt) {
        const int k = (kb0 + e) * 16 + 4 * q, n = (nt0 + f) * 16 + i;
        if (P.transpose_out) {
          *reinterpret_cast<acc_t*>(out + (size_t)n * P.ld_out + k) = acc[e][f];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) out[(size_t)(k + r) * P.ld_out + n] = acc[e][f][r];
        }
      }
    }
  }
  if (a.dbg && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {
    ts3 = __builtin_amdgcn_s_memtime();
    unsigned long long* d = a.dbg + (blockIdx.x == 0 ? 0 : 8);
    d[0] = ts0; d[1] = ts1; d[2] = tsm; d[3] = ts2; d[4] = ts3;
  }
}

template <int KTW, int NTW, int CH>
__global__ __launch_bounds__(256, 2) void gemm_tn_multi_kernel(TnMulti a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char goctr_smem[];
  int pi = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k) if (k < a.np && (int)blockIdx.x >= a.p[k].first) pi = k;
  const TnProblem& P = a.p[pi];
  const int local = blockIdx.x - P.first;
  const int kb = local / P.S, split = local - kb * P.S;
  if (P.KT == 1) tn_multi_body<1, NTW, CH>(a, P, kb, split, goctr_smem);   // one-tile problems: a third of the MFMAs
  else tn_multi_body<KTW, NTW, CH>(a, P, kb, split, goctr_smem);
}
// LDS bytes for problems whose widest D operand has nt_max 16-column tiles
template <int KTW, int NTW, int CH>
inline size_t gemm_tn_multi_lds_bytes(int nt_max) {
  return sizeof(float) * 2 * (CH + 4) * (size_t)(KTW * 16 + nt_max * 16);
}
template <int KTW, int CH>
inline bool gemm_tn_multi_fits(int nt_max) { return nt_max <= 16; }

// ---------------------------------------------------------------------------------------------------------------------
// The same weight-gradient kernel on the 6-product bf16 split (scripts/ubench/bf16x3.hip): every float32
// operand value x is staged as three bf16 planes  x = hi + mid + lo  (each the round-to-nearest-even bf16 of what the
// previous ones left), and  a*b  is taken as  hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid)  on
// v_mfma_f32_16x16x32_bf16 with float32 accumulation — the dropped terms are below 2^-32 |a b|.  Measured against float64
// it is MORE accurate than v_mfma_f32_16x16x4_f32 on this path's operands (2.8e-8 vs 1.3e-7 relative to sum |a b|), it
// issues 1.4x faster, and unlike the f32-input MFMA it does not block the SIMD for the other wavefronts' staging work.
// One MFMA covers the whole 32-row chunk (k-depth 32): per chunk and tile pair 6 MFMAs instead of 8.
// LDS: per column and plane CHB = 40 bf16 (80 B: the 16-byte reads of 16 lanes fall in distinct 16-byte bank groups).
typedef __bf16 tn_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 tn_bf4 __attribute__((ext_vector_type(4)));

typedef __bf16 tn_bf2 __attribute__((ext_vector_type(2)));
typedef float tn_f2 __attribute__((ext_vector_type(2)));
// two values at a time: v_cvt_pk_bf16_f32 rounds both (RNE) and leaves them packed the way the LDS strips want them
// (consecutive rows adjacent); widening a packed bf16 back to float32 is a shift / a mask.
__device__ __forceinline__ void tn_split3_pk(float x0, float x1, unsigned int& hi, unsigned int& mid, unsigned int& lo) {
  hi = __builtin_bit_cast(unsigned int, __builtin_convertvector(tn_f2{x0, x1}, tn_bf2));
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xFFFF0000u);
  mid = __builtin_bit_cast(unsigned int, __builtin_convertvector(tn_f2{r0, r1}, tn_bf2));
  const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xFFFF0000u);
  lo = __builtin_bit_cast(unsigned int, __builtin_convertvector(tn_f2{s0, s1}, tn_bf2));
}

// (n0t, nb_t): the block's D tiles [n0t, n0t + nb_t), nb_t <= 4 * NTW.  KTW > 3 ("wide" blocks, round 2): the A fragments
// of one tile at a time stream through registers (KTW * NTW * 8 accumulator registers leave no room for all of them).
template <int KTW, int NTW>
__device__ __forceinline__ void tn_multi_body_x3(const TnMulti& a, const TnProblem& P, int kb, int n0t, int nb_t, int split,
                                                 unsigned char* smem) {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  typedef float vec_t __attribute__((ext_vector_type(4)));
  constexpr int CH = 32, CHB = 40;        // rows per chunk; bf16 per LDS strip column
  constexpr int MAXB = ((CH / 4) * (KTW * 4 + 4 * NTW * 4) + 255) / 256;
  const int kb0 = kb * KTW;
  int kb_t = P.KT - kb0; if (kb_t > KTW) kb_t = KTW;
  const int Kc = kb_t * 16, Nc = nb_t * 16;
  const int kv = Kc >> 2, nv = Nc >> 2;
  // LDS (bf16 units): [buf][plane][col][CHB], A columns first (KTW*16 of them), then the D columns
  __bf16* base = reinterpret_cast<__bf16*>(smem);
  const int cols = KTW * 16 + Nc;
  const int plane = cols * CHB, bufsz = 3 * plane;

  // 8 wavefronts: 0-3 multiply (NTW tiles of D each), 4-7 stage — one of each kind per SIMD, so that the bf16 MFMAs of
  // the one run under the float32 -> 3 x bf16 splitting, global loads and LDS writes of the other.
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m_begin = split * P.rows;
  int m_end = m_begin + P.rows;
  if (m_end > a.M) m_end = a.M;
  if (m_begin >= m_end) return;            // (workgroup-uniform)
  const int nchunks = (m_end - m_begin + CH - 1) / CH;

  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  const int nA = (CH / 4) * kv, nAll = nA + (CH / 4) * nv;
  // block b = (column group cg, row group r) with r fastest: the 8 lanes of a column group write 64 contiguous bytes of a
  // strip (cg-fastest put 16 lanes on the same LDS banks), and a wavefront's loads still touch 16 cache lines
  // The problem's operand pointers and row strides as SCALARS, pinned: left to the compiler, "stride = block of A ? P.lda : P.ldd"
  // became ONE vector load from the selected ADDRESS inside the kernel-argument buffer (and the same for the pointer) -- a memory
  // round trip per lane in front of the first row request of every staging pass; the launch's "wait for chunk 0" (GOCTR_DBG=tn) was
  // two dependent round trips, descriptor then rows.
  // (pinned as integers and re-typed as GLOBAL pointers: a pointer that went through the asm statement is a generic one to the
  // compiler, and its loads would be flat_load -- counted on the LDS counter too)
  typedef const float __attribute__((address_space(1)))* tn_gptr;
  unsigned long long pa_u = reinterpret_cast<unsigned long long>(P.A), pd_u = reinterpret_cast<unsigned long long>(P.D);
  int plda = P.lda, pldd = P.ldd;
  asm volatile("" : "+s"(pa_u), "+s"(pd_u), "+s"(plda), "+s"(pldd));
  const tn_gptr PA = (tn_gptr)pa_u, PD = (tn_gptr)pd_u;
  auto slot_of = [&](int b, tn_gptr& src, int& ldv, int& rgv, int& lo) {
    src = PA; ldv = plda; rgv = 0; lo = -1;
    if (b < nA) {
      const int cg = b >> 3, r = b & 7;
      rgv = r; ldv = plda;
      src = PA + kb0 * 16 + cg * 4;
      lo = (cg * 4) * CHB + 4 * r;
    } else if (b < nAll) {
      const int bb = b - nA;
      const int cg = bb >> 3, r = bb & 7;
      rgv = r; ldv = pldd;
      src = PD + n0t * 16 + cg * 4;
      lo = (KTW * 16 + cg * 4) * CHB + 4 * r;
    }
  };
  auto store_block = [&](__bf16* d, int left, const vec_t (&v)[4]) {     // 4 rows x 4 columns -> 3 planes, zero past `left` rows
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x0 = left > 0 ? v[0][c] : 0.f, x1 = left > 1 ? v[1][c] : 0.f;
      const float x2 = left > 2 ? v[2][c] : 0.f, x3 = left > 3 ? v[3][c] : 0.f;
      unsigned int h0, m0_, l0, h1, m1, l1;
      tn_split3_pk(x0, x1, h0, m0_, l0);
      tn_split3_pk(x2, x3, h1, m1, l1);
      *reinterpret_cast<u2*>(d + c * CHB) = u2{h0, h1};
      *reinterpret_cast<u2*>(d + plane + c * CHB) = u2{m0_, m1};
      *reinterpret_cast<u2*>(d + 2 * plane + c * CHB) = u2{l0, l1};
    }
  };
  // chunk 0 -> buffer 0 by all 512 threads (the multipliers have nothing else to do yet)
  auto stage_first = [&]() {
    for (int b = tid; b < nAll; b += 512) {
      tn_gptr src; int ldv, rgv, lo;
      slot_of(b, src, ldv, rgv, lo);
      vec_t v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int rr = m_begin + 4 * rgv + r;
        rr = rr < m_end ? rr : m_end - 1;
        v[r] = *reinterpret_cast<const vec_t __attribute__((address_space(1)))*>(src + (size_t)rr * ldv);
      }
      store_block(base + lo, m_end - (m_begin + 4 * rgv), v);
    }
  };

  if (wave >= 4) {
    // ------------------------------------------------------------------------------------------------ stagers
    const int stid = tid - 256;
    tn_gptr gsrc[MAXB]; int ld[MAXB]; int rg[MAXB]; int lofs[MAXB];   // lofs: bf16 offset inside a plane; < 0 unused
#pragma unroll
    for (int s = 0; s < MAXB; ++s) slot_of(stid + s * 256, gsrc[s], ld[s], rg[s], lofs[s]);
    // loads are unconditional (rows past the slab's end re-read its last row and are zeroed at the LDS write)
    auto gload = [&](int m0, vec_t (&st)[MAXB][4]) {
#pragma unroll
      for (int s = 0; s < MAXB; ++s) {
        if (s * 256 < nAll) {                                // (uniform: slots past the problem's block count are not loaded)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int rr = m0 + 4 * rg[s] + r;
            rr = rr < m_end ? rr : m_end - 1;
            st[s][r] = *reinterpret_cast<const vec_t __attribute__((address_space(1)))*>(gsrc[s] + (size_t)rr * ld[s]);
          }
        }
      }
    };
    auto lstore = [&](int buf, int m0, const vec_t (&st)[MAXB][4]) {
      const bool whole = m0 + CH <= m_end;                   // whole chunk inside the slab (uniform): nothing to zero
#pragma unroll
      for (int s = 0; s < MAXB; ++s)
        if (s * 256 < nAll && lofs[s] >= 0)
          store_block(base + buf * bufsz + lofs[s], whole ? 4 : m_end - (m0 + 4 * rg[s]), st[s]);
    };
    // chunk c is written while chunk c - 1 is multiplied; its global loads were issued one iteration earlier (two ahead
    // was measured: the extra loads only lengthen the start).  Chunk 0 was staged by all 8 wavefronts (below).
    unsigned long long tS0 = 0, tS1 = 0, tSt = 0, tS2 = 0;
    if (a.dbg) tS0 = __builtin_amdgcn_s_memtime();
    vec_t st0[MAXB][4], st1[MAXB][4];
    if (nchunks > 1) gload(m_begin + CH, st1);
    stage_first();
    __syncthreads();                                       // chunk 0 is in LDS
    if (a.dbg) tS1 = __builtin_amdgcn_s_memtime();
    for (int c = 1; c <= nchunks; ++c) {
      unsigned long long t0 = 0;
      if (a.dbg) t0 = __builtin_amdgcn_s_memtime();
      if (c < nchunks) {
#pragma unroll
        for (int s = 0; s < MAXB; ++s)
#pragma unroll
          for (int r = 0; r < 4; ++r) st0[s][r] = st1[s][r];
        if (c + 1 < nchunks) gload(m_begin + (c + 1) * CH, st1);
        lstore(c & 1, m_begin + c * CH, st0);
      }
      if (a.dbg) tSt += __builtin_amdgcn_s_memtime() - t0;
      __syncthreads();                                     // chunk c written, chunk c - 1 consumed
    }
    if (a.dbg && tid == 256 && blockIdx.x == 0) {
      tS2 = __builtin_amdgcn_s_memtime();
      a.dbg[8] = tS1 - tS0; a.dbg[9] = tSt; a.dbg[10] = tS2 - tS0; a.dbg[11] = nchunks;
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- multipliers
  const int wn = wave;
  const int i = lane & 15, q = lane >> 4;
  const int nt0 = wn * NTW;
  int ncnt = nb_t - nt0; ncnt = ncnt < 0 ? 0 : (ncnt > NTW ? NTW : ncnt);
  acc_t ah[KTW][NTW], ac[KTW][NTW];       // hi*hi, and everything else
#pragma unroll
  for (int e = 0; e < KTW; ++e)
#pragma unroll
    for (int f = 0; f < NTW; ++f) { ah[e][f] = acc_t{0, 0, 0, 0}; ac[e][f] = acc_t{0, 0, 0, 0}; }
  int aofs[KTW], dofs[NTW];
#pragma unroll
  for (int e = 0; e < KTW; ++e) { int c = e * 16 + i; c = c < Kc ? c : Kc - 1; aofs[e] = c * CHB + 8 * q; }
#pragma unroll
  for (int f = 0; f < NTW; ++f) { int c = (nt0 + f) * 16 + i; c = c < Nc ? c : Nc - 1; dofs[f] = (KTW * 16 + c) * CHB + 8 * q; }

  unsigned long long tC0 = 0, tC1 = 0, tCm = 0, tC2 = 0;
  if (a.dbg) tC0 = __builtin_amdgcn_s_memtime();
  stage_first();
  __syncthreads();                                         // chunk 0 is in LDS
  if (a.dbg) tC1 = __builtin_amdgcn_s_memtime();
  for (int c = 0; c < nchunks; ++c) {
    unsigned long long t0 = 0;
    if (a.dbg) t0 = __builtin_amdgcn_s_memtime();
    const __bf16* bs = base + (c & 1) * bufsz;
    if constexpr (KTW <= 3) {
      tn_bf8 av[3][KTW], dv[3][NTW];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int e = 0; e < KTW; ++e) av[pl][e] = *reinterpret_cast<const tn_bf8*>(bs + pl * plane + aofs[e]);
#pragma unroll
        for (int f = 0; f < NTW; ++f) dv[pl][f] = *reinterpret_cast<const tn_bf8*>(bs + pl * plane + dofs[f]);
      }
#pragma unroll
      for (int e = 0; e < KTW; ++e)
#pragma unroll
        for (int f = 0; f < NTW; ++f) {
          // smallest terms first inside the correction accumulator
          ac[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[2][e], dv[0][f], ac[e][f], 0, 0, 0);
          ac[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0][e], dv[2][f], ac[e][f], 0, 0, 0);
          ac[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1][e], dv[1][f], ac[e][f], 0, 0, 0);
          ac[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1][e], dv[0][f], ac[e][f], 0, 0, 0);
          ac[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0][e], dv[1][f], ac[e][f], 0, 0, 0);
          ah[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0][e], dv[0][f], ah[e][f], 0, 0, 0);
        }
    } else {
      tn_bf8 dv[3][NTW], av[3], an[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int f = 0; f < NTW; ++f) dv[pl][f] = *reinterpret_cast<const tn_bf8*>(bs + pl * plane + dofs[f]);
        an[pl] = *reinterpret_cast<const tn_bf8*>(bs + pl * plane + aofs[0]);
      }
#pragma unroll
      for (int e = 0; e < KTW; ++e) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          av[pl] = an[pl];
          if (e + 1 < KTW) an[pl] = *reinterpret_cast<const tn_bf8*>(bs + pl * plane + aofs[e + 1]);   // the next tile's fragments
        }
#pragma unroll
        for (int f = 0; f < NTW; ++f) {
          ac[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[2], dv[0][f], ac[e][f], 0, 0, 0);
          ac[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], dv[2][f], ac[e][f], 0, 0, 0);
          ac[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], dv[1][f], ac[e][f], 0, 0, 0);
          ac[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[1], dv[0][f], ac[e][f], 0, 0, 0);
          ac[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], dv[1][f], ac[e][f], 0, 0, 0);
          ah[e][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[0], dv[0][f], ah[e][f], 0, 0, 0);
        }
      }
    }
    if (a.dbg) tCm += __builtin_amdgcn_s_memtime() - t0;
    __syncthreads();                                       // chunk c consumed, chunk c + 1 written
  }
  if (a.dbg) tC2 = __builtin_amdgcn_s_memtime();

  // slab stores: wave-uniform base + 32-bit element offsets (a slab is < 2^31 bytes): one integer add per store instead of
  // a 64-bit multiply-add chain (the epilogue of a 9 x 2-tile wavefront took 5.1 k cycles of address arithmetic)
  float* out = P.slabs + (size_t)split * P.slab_stride;
  const unsigned ldo = (unsigned)P.ld_out;
  const unsigned k0 = (unsigned)(kb0 * 16 + 4 * q), n0 = (unsigned)((n0t + nt0) * 16 + i);
  const unsigned o_n = k0 * ldo + n0, o_t = n0 * ldo + k0;
#pragma unroll
  for (int e = 0; e < KTW; ++e) {
#pragma unroll
    for (int f = 0; f < NTW; ++f) {
      if (e < kb_t && f < ncnt) {
        const acc_t v = ac[e][f] + ah[e][f];
        if (P.transpose_out) {
          const unsigned o = o_t + (unsigned)(f * 16) * ldo + (unsigned)(e * 16);
          if (a.wt) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" : : "v"(o * 4u), "v"(v), "s"(out) : "memory");
          else *reinterpret_cast<acc_t*>(out + o) = v;
        } else {
          const unsigned o = o_n + (unsigned)(e * 16) * ldo + (unsigned)(f * 16);
          if (a.wt >= 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("global_store_dword %0, %1, %2 sc1" : : "v"((o + (unsigned)r * ldo) * 4u), "v"(v[r]), "s"(out) : "memory");
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[o + (unsigned)r * ldo] = v[r];
          }
        }
      }
    }
  }
  if (a.dbg && tid == 0 && blockIdx.x == 0) {
    a.dbg[0] = tC1 - tC0; a.dbg[1] = tCm; a.dbg[2] = tC2 - tC1; a.dbg[3] = __builtin_amdgcn_s_memtime() - tC2;
  }
}

template <int KTW, int NTW>
__global__ __launch_bounds__(512) void gemm_tn_multi_x3_kernel(TnMulti a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char goctr_smem[];
  int pi = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k) if (k < a.np && (int)blockIdx.x >= a.p[k].first) pi = k;
  const TnProblem& P = a.p[pi];
  const int local = blockIdx.x - P.first;
  const int kb = local / P.S, split = local - kb * P.S;
  if (P.KT == 1) tn_multi_body_x3<1, NTW>(a, P, kb, 0, P.NT, split, goctr_smem);
  else tn_multi_body_x3<KTW, NTW>(a, P, kb, 0, P.NT, split, goctr_smem);
}

// Round 6: "sum problems" (TnProblem::KT == 0).  Where the chain launch has already summed a gradient over the 32 rows of each of
// its tiles (ctr_chain_x3.h: dW2 = A1^T dz2 and the att0 terms -- their operands A1, dz2 and the per-sample terms then never
// leave the chip, 5.2 MB less written by the chain launch and read by this one at cfg3), what is left is a sum over the TILES:
// D = the partials [tiles][lda], workgroup `split` adds tiles [split * rows, (split + 1) * rows) in ascending order (16 loads in
// flight) and writes slab `split` in the layout the MFMA problem wrote: transposed (dW2): out[idx * ld_out + 0] with the other
// ld_out - 1 entries of the row zero; untransposed (att0): out[idx] (row 0 of the slab, the only one the reduce reads).
__device__ __forceinline__ void tn_tile_sum_body(const TnMulti& a, const TnProblem& P, int split) {
  const int ntiles = (a.M + 31) >> 5;
  const int t0 = split * P.rows;
  int t1 = t0 + P.rows; if (t1 > ntiles) t1 = ntiles;
  float* out = P.slabs + (size_t)split * P.slab_stride;
  for (int idx = threadIdx.x; idx < P.lda; idx += blockDim.x) {
    float s = 0.f;
    for (int tb = t0; tb < t1; tb += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int t = tb + u < t1 ? tb + u : t1 - 1; v[u] = P.D[(size_t)t * P.ldd + idx]; }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += tb + u < t1 ? v[u] : 0.f;
    }
    if (P.transpose_out) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      float* o = out + (size_t)idx * P.ld_out;
      *reinterpret_cast<f4*>(o) = f4{s, 0.f, 0.f, 0.f};
      for (int k = 4; k < P.ld_out; k += 4) *reinterpret_cast<f4*>(o + k) = f4{0.f, 0.f, 0.f, 0.f};
    } else {
      out[idx] = s;
    }
  }
}

// Round 2: wide blocks.  Problem 0 (dW0) takes ALL its A tiles (KTW0 = 9 DIN / 8 per k-block YouTube) and problem 1 (dW1,
// posed transposed) all its KTW1 = 5 against HALF of the D tiles (two n-blocks of <= 8 tiles, 2 per multiplying wavefront):
// per slab the workgroups stage 496 + 368 operand columns instead of 768 + 496 (every D column was staged once per
// 3-tile k-block: dz0 three times, A0 twice), and the slabs of the two problems get their own heights so that the
// differently sized blocks cost the same.  One-tile problems keep the <1, 4> body.
// (the kernel's body as a function: ctr_chain_x3.h's gemm_tn_multi_x3w_att0_kernel runs it in all of its workgroups but the last)
template <int KTW0, int KTW1>
__device__ __forceinline__ void tn_multi_x3w_block(const TnMulti& a, unsigned char* goctr_smem) {
  int pi = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k) if (k < a.np && (int)blockIdx.x >= a.p[k].first) pi = k;
  const TnProblem& P = a.p[pi];
  const int local = blockIdx.x - P.first;
  const int blk = local / P.S, split = local - blk * P.S;
  if (P.KT == 0) { tn_tile_sum_body(a, P, local); return; }
  if (P.KT == 1) { tn_multi_body_x3<1, 4>(a, P, blk, 0, P.NT, split, goctr_smem); return; }
  const int nnb = P.nnb > 0 ? P.nnb : 1;
  const int kb = blk / nnb, nb = blk - kb * nnb;
  const int n0t = nb * P.nbt;
  int nb_t = P.NT - n0t; if (nb_t > P.nbt) nb_t = P.nbt;
  if (pi == 0) tn_multi_body_x3<KTW0, 2>(a, P, kb, n0t, nb_t, split, goctr_smem);
  else tn_multi_body_x3<KTW1, 2>(a, P, kb, n0t, nb_t, split, goctr_smem);
}
template <int KTW0, int KTW1>
__global__ __launch_bounds__(512) void gemm_tn_multi_x3w_kernel(TnMulti a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char goctr_smem[];
  tn_multi_x3w_block<KTW0, KTW1>(a, goctr_smem);
}
template <int KTW>
inline size_t gemm_tn_multi_x3w_lds_bytes() {
  return sizeof(__bf16) * 2 * 3 * 40 * (size_t)(KTW * 16 + 8 * 16);
}
template <int KTW>
inline size_t gemm_tn_multi_x3_lds_bytes(int nt_max) {
  return sizeof(__bf16) * 2 * 3 * 40 * (size_t)(KTW * 16 + nt_max * 16);
}

}  // namespace goctr
