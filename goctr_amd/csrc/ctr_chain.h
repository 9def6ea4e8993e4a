// ctr_chain.h -- the fused per-row-tile chain of the DIN / YouTube step (float32):
//
//   h0 -> sigma(h0.W0) -> drop -> sigma(.W1) -> drop -> sigma(.W2) -> BCE term, dz2 -> dz1 -> dz0 -> dp
//   (model/din/din.go:301-315, model/cost.go:9-17 and their hand-derived backward)
//
// in ONE launch instead of six GEMM launches with five activation round trips through HBM.
//
// Register-resident activations: every product is computed TRANSPOSED,
//   Z^T[n][row] = sum_k W[k][n] * X[row][k]      (MFMA A-operand = W tile from LDS, B-operand = X),
// so the v_mfma_f32_16x16x4_f32 accumulator of lane (row = lane&15, q = lane>>4) holds
// Z[row][16t + 4q + r], r = 0..3 -- exactly the B-operand fragment (4 consecutive, k-permuted values per lane)
// the NEXT layer's MFMAs want.  A wavefront carries its 16 batch rows through all layers without touching
// LDS or shuffling; LDS only stages the weight operands shared by the workgroup's 4 wavefronts.
//
// LDS weight image: [k/4][n][k%4] -- the 4 k-values one lane feeds to 4 consecutive MFMAs are one 16-byte
// ds_read_b128 (measured: one ds_read_b32 per MFMA costs 47 cycles/MFMA against 32 for register operands,
// scripts/ubench/mfma_rate.hip).  The image is produced while staging: a thread loads a 4x4 block (4 coalesced
// 16-byte global loads), transposes it in registers and writes 4 ds_write_b128.  Rows of n are 16 B apart, so
// the 16 lanes of a q-group read one contiguous 256 B bank row: conflict-free.
//
// Latency hiding with one wavefront per SIMD: the global loads of the NEXT weight operand are issued into
// registers before the current MFMA phase and written to LDS after it (the staging area is single-buffered).
//
// Workgroup = 4 wavefronts = 2 row tiles (32 batch rows) x 2 halves: the two wavefronts of a row tile split
// the H1 = 208 columns 7 + 6 tiles, so each holds half of A0 -- i.e. half of the K range of the next GEMM --
// and the partial Z1 (and later partial dp) are summed through a 20 KB LDS exchange.  8192 rows => 256
// workgroups => one per CU, 1024 wavefronts => one per SIMD.
#pragma once
#include <hip/hip_runtime.h>

#include "ctr_kernels.h"
#include "mfma_gemm.h"

namespace goctr {

constexpr int CHAIN_KPH0 = 80;   // rows of W0 per LDS phase (5 chunks of 16)
constexpr int CHAIN_NDP = 4;     // max 16-wide tiles of the pooled-embedding gradient (D <= 64)
constexpr int CHAIN_HV = 10;     // h0 fragments (16 columns each) a lane keeps in registers: Ip <= 160
constexpr int CHAIN_PF = 5;      // 4x4 blocks a thread stages per operand (5*256 >= 1248 blocks)

struct ChainArgs {
  const float* h0; int Ip;                       // [B, Ip]
  const float* W0; const float* W1; const float* W2;   // padded [Ip,H1p], [H1p,H2p], [H2p,16]
  const float* W1T; const float* W0sT;                 // [H2p,H1p], [H1p,Dp]
  int H1, H2, H1p, H2p, Dp; int B;
  int train; int kind;
  DropCfg d0, d1; const StepState* st;
  const float* Y; long long rows; float inv_bglobal;
  int wb_floats;                                 // size of the weight staging area (floats)
  // outputs (training only, except yhat)
  float* A0; float* A1; float* dz0; float* dz1; float* dz2; float* dp; float* yhat; float* lossrow;
  unsigned long long* dbg;                       // optional [16] phase timestamps of block 0 / wave 0 (s_memtime)
};

inline int chain_wb_floats(int Ip, int H1p, int H2p, int Dp) {
  const int kph = Ip < CHAIN_KPH0 ? Ip : CHAIN_KPH0;
  int a = kph * H1p;
  int b = H1p * H2p;
  int c = H2p * H1p + H1p * Dp;
  int m = a > b ? a : b;
  return (m > c ? m : c) + 256;  // slack: edge waves read (and discard) one tile past the block
}
template <int NT1>
inline size_t chain_lds_bytes(int Ip, int H1p, int H2p, int Dp) {
  return sizeof(float) * ((size_t)chain_wb_floats(Ip, H1p, H2p, Dp) + 4 * 64 * (NT1 * 4 > CHAIN_NDP * 4 ? NT1 * 4 : CHAIN_NDP * 4));
}

typedef float chain_f4 __attribute__((ext_vector_type(4)));

// One staged operand = a [K x N] row-major f32 block in global memory (K % 4 == 0, N % 4 == 0).  A thread
// owns up to CHAIN_PF 4x4 blocks: load = 4 coalesced 16-byte loads per block (issued early), store = the
// transposed block as 4 ds_write_b128 into the [K/4][N][4] LDS image.
struct ChainPf { chain_f4 v[CHAIN_PF][4]; };

__device__ __forceinline__ void chain_pf_load(ChainPf& pf, const float* __restrict__ src, int src_ld, int K, int N,
                                              int tid, int blk0 = 0) {
  const int nb = N >> 2, total = (K >> 2) * nb;
  const float rnb = 1.0f / (float)nb;
#pragma unroll
  for (int s = 0; s < CHAIN_PF; ++s) {
    const int b = tid + s * 256 - blk0;
    if (b >= 0 && b < total) {
      const int kq = (int)(((float)b + 0.5f) * rnb), n4 = b - kq * nb;   // exact for b < 2^20
      const float* p = src + (size_t)(4 * kq) * src_ld + 4 * n4;
#pragma unroll
      for (int r = 0; r < 4; ++r) pf.v[s][r] = *reinterpret_cast<const chain_f4*>(p + (size_t)r * src_ld);
    }
  }
}
__device__ __forceinline__ void chain_pf_store(const ChainPf& pf, float* dst, int K, int N, int tid, int blk0 = 0) {
  const int nb = N >> 2, total = (K >> 2) * nb;
  const float rnb = 1.0f / (float)nb;
#pragma unroll
  for (int s = 0; s < CHAIN_PF; ++s) {
    const int b = tid + s * 256 - blk0;
    if (b >= 0 && b < total) {
      const int kq = (int)(((float)b + 0.5f) * rnb), n4 = b - kq * nb;
      float* d = dst + ((size_t)kq * N + 4 * n4) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<chain_f4*>(d + 4 * j) = chain_f4{pf.v[s][0][j], pf.v[s][1][j], pf.v[s][2][j], pf.v[s][3][j]};
    }
  }
}

// dropout scale factor mask/keep for (row, col); DROP: 0 none, 1 explicit mask, 2 counter hash
template <int DROP>
__device__ __forceinline__ float chain_dropk(const DropCfg& d, const StepState* st, int row, int col) {
  if (DROP == 0 || !d.mode) return 1.0f;
  const float keep = 1.0f - d.p;
  float m;
  if (DROP == 1) m = d.mask[(size_t)row * d.mask_ld + col];
  else m = dropout_keep(d.seed, st->gstep, d.layer, d.row_off + row, col, d.p);
  return m / keep;
}

// branch-free variant of sigm_hidden
__device__ __forceinline__ float chain_sigm(float x) {
  float s = __builtin_amdgcn_rcpf(1.0f + __expf(-x));
  s = x > 15.f ? 1.0f : s;
  return x < -88.f ? 0.0f : s;
}

template <int NT0H, int NT1, int DROP>
__global__ __launch_bounds__(256, 1) void ctr_chain_kernel(ChainArgs a) {
  typedef chain_f4 f4;
  using MF = Mfma<float>;
  extern __shared__ __attribute__((aligned(16))) float chain_smem[];
  float* Wb = chain_smem;
  float* xch = chain_smem + a.wb_floats;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rt = wave >> 1, hf = wave & 1;
  const int i = lane & 15, q = lane >> 4;
  const int row = blockIdx.x * 32 + rt * 16 + i;
  const bool vrow = row < a.B;
  const int rowc = vrow ? row : a.B - 1;
  const int H1p = a.H1p, H2p = a.H2p, Ip = a.Ip, Dp = a.Dp;
  const int NT0 = H1p >> 4;
  const int t0 = hf * NT0H;
  int ntl = NT0 - t0;
  ntl = ntl > NT0H ? NT0H : (ntl < 0 ? 0 : ntl);
  const bool full = ntl == NT0H;   // the second half of a 13-tile layer owns one tile less
  int stamp_i = 0;
  auto stamp = [&]() {
    if (a.dbg && blockIdx.x == 0 && tid == 0) a.dbg[stamp_i] = __builtin_amdgcn_s_memtime();
    ++stamp_i;
  };
  stamp();  // 0

  // ------------------------------------------------------------------ F0: Z0^T = W0^T . h0^T
  f4 acc0[NT0H];
#pragma unroll
  for (int t = 0; t < NT0H; ++t) acc0[t] = f4{0, 0, 0, 0};
  const float* hp = a.h0 + (size_t)rowc * Ip + 4 * q;
  // every h0 fragment of this lane: one burst of 16-byte loads up front (host guarantees Ip <= 16*CHAIN_HV)
  f4 hall[CHAIN_HV];
#pragma unroll
  for (int c = 0; c < CHAIN_HV; ++c)
    if (c * 16 < Ip) hall[c] = *reinterpret_cast<const f4*>(hp + c * 16);
  ChainPf pf;
  {
    const int kph = Ip < CHAIN_KPH0 ? Ip : CHAIN_KPH0;
    chain_pf_load(pf, a.W0, H1p, kph, H1p, tid);
  }
  for (int k0 = 0; k0 < Ip; k0 += CHAIN_KPH0) {
    const int kph = Ip - k0 < CHAIN_KPH0 ? Ip - k0 : CHAIN_KPH0;
    const int nch = kph >> 4;
    if (k0 > 0) __syncthreads();          // previous phase's LDS reads are done
    chain_pf_store(pf, Wb, kph, H1p, tid);
    __syncthreads();
    stamp();  // 1 (3): W0 phase in LDS
    // prefetch the next operand while this phase multiplies
    const int k1 = k0 + CHAIN_KPH0;
    if (k1 < Ip) chain_pf_load(pf, a.W0 + (size_t)k1 * H1p, H1p, Ip - k1 < CHAIN_KPH0 ? Ip - k1 : CHAIN_KPH0, H1p, tid);
    else chain_pf_load(pf, a.W1, H2p, H1p, H2p, tid);
    const float* wp = Wb + ((size_t)q * H1p + t0 * 16 + i) * 4;
    f4 hcur[CHAIN_KPH0 / 16];
#pragma unroll
    for (int c = 0; c < CHAIN_KPH0 / 16; ++c) hcur[c] = k0 == 0 ? hall[c] : hall[c + CHAIN_KPH0 / 16 < CHAIN_HV ? c + CHAIN_KPH0 / 16 : CHAIN_HV - 1];
#pragma unroll
    for (int c = 0; c < CHAIN_KPH0 / 16; ++c) {
      if (c < nch) {
        f4 w4[NT0H];
#pragma unroll
        for (int t = 0; t < NT0H; ++t) w4[t] = *reinterpret_cast<const f4*>(wp + t * 64);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < NT0H; ++t)
            if (t < NT0H - 1 || full) acc0[t] = MF::mma(w4[t][r], hcur[c][r], acc0[t]);
        wp += (size_t)4 * H1p * 4;
      }
    }
    stamp();  // 2 (4): phase MFMAs issued
  }
  // sigmoid + dropout in registers; acc0 becomes A0 (post-dropout), p0 keeps the pre-dropout sigmoid
  f4 p0[NT0H];
#pragma unroll
  for (int t = 0; t < NT0H; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = (t0 + t) * 16 + 4 * q + r;
      float s = chain_sigm(acc0[t][r]);
      s = (n < a.H1 && (t < NT0H - 1 || full)) ? s : 0.0f;
      p0[t][r] = s;
      acc0[t][r] = DROP ? s * chain_dropk<DROP>(a.d0, a.st, row, n) : s;
    }
    if (a.train && vrow && (t < NT0H - 1 || full))
      *reinterpret_cast<f4*>(a.A0 + (size_t)row * H1p + (t0 + t) * 16 + 4 * q) = acc0[t];
  }
  stamp();  // 5: F0 epilogue

  // ------------------------------------------------------------------ F1: Z1^T = W1^T . A0^T (K split over the pair)
  __syncthreads();
  chain_pf_store(pf, Wb, H1p, H2p, tid);
  __syncthreads();
  stamp();  // 6: W1 in LDS
  if (a.train) {  // prefetch W1^T and (DIN) W0[U:U+D,:]^T for the backward phases
    chain_pf_load(pf, a.W1T, H1p, H2p, H1p, tid);
    if (a.kind == GOCTR_DIN) chain_pf_load(pf, a.W0sT, Dp, H1p, Dp, tid, (H2p >> 2) * (H1p >> 2));
  }
  f4 acc1[NT1];
#pragma unroll
  for (int u = 0; u < NT1; ++u) acc1[u] = f4{0, 0, 0, 0};
  {
    const float* wp = Wb + ((size_t)(t0 * 4 + q) * H2p + i) * 4;
#pragma unroll
    for (int t = 0; t < NT0H; ++t) {
      if (t < NT0H - 1 || full) {
        f4 w4[NT1];
#pragma unroll
        for (int u = 0; u < NT1; ++u) w4[u] = *reinterpret_cast<const f4*>(wp + u * 64);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int u = 0; u < NT1; ++u) acc1[u] = MF::mma(w4[u][r], acc0[t][r], acc1[u]);
        wp += (size_t)4 * H2p * 4;
      }
    }
  }
  stamp();  // 7: F1 MFMAs
  constexpr int XS = NT1 * 4 > CHAIN_NDP * 4 ? NT1 * 4 : CHAIN_NDP * 4;
  {
    float* xw = xch + ((rt * 2 + hf) * XS) * 64 + lane;
#pragma unroll
    for (int u = 0; u < NT1; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) xw[(u * 4 + r) * 64] = acc1[u][r];
    __syncthreads();
    const float* xr = xch + ((rt * 2 + (hf ^ 1)) * XS) * 64 + lane;
#pragma unroll
    for (int u = 0; u < NT1; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc1[u][r] += xr[(u * 4 + r) * 64];  // a+b == b+a: both halves agree bitwise
  }
  f4 p1[NT1];
  f4 w2v[NT1];
  float part = 0.f;
#pragma unroll
  for (int u = 0; u < NT1; ++u) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = u * 16 + 4 * q + r;
      float s = chain_sigm(acc1[u][r]);
      s = n < a.H2 ? s : 0.0f;
      p1[u][r] = s;
      const float post = DROP ? s * chain_dropk<DROP>(a.d1, a.st, row, n) : s;
      acc1[u][r] = post;
      w2v[u][r] = a.W2[(size_t)n * 16];
      part += post * w2v[u][r];
    }
    if (a.train && vrow && hf == 0) *reinterpret_cast<f4*>(a.A1 + (size_t)row * H2p + u * 16 + 4 * q) = acc1[u];
  }
  // ------------------------------------------------------------------ output unit, BCE term, dz2, dz1
  float z2 = part + __shfl_xor(part, 16, 64);
  z2 += __shfl_xor(z2, 32, 64);
  const float yh = sigm_out(z2);
  const bool writer = hf == 0 && q == 0 && vrow;
  if (writer) a.yhat[row] = yh;
  if (!a.train) return;
  const long long gr = a.st->batch_idx * (long long)a.B + row;
  const float y = (a.Y && vrow && gr < a.rows) ? a.Y[gr] : 0.f;
  const float one_eps = (float)(1.0 + 1e-8);
  const float dy = -((y / yh) - ((1.0f - y) / (one_eps - yh))) * a.inv_bglobal;
  const float d2 = dy * (yh * (1.0f - yh));
  if (writer) {
    a.lossrow[row] = logf(yh) * y + logf(one_eps - yh) * (1.0f - y);
    a.dz2[(size_t)row * 16] = d2;
  }
#pragma unroll
  for (int u = 0; u < NT1; ++u) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = u * 16 + 4 * q + r;
      const float k = DROP ? chain_dropk<DROP>(a.d1, a.st, row, n) : 1.0f;
      const float s = p1[u][r];
      acc1[u][r] = ((d2 * w2v[u][r]) * k) * (s * (1.0f - s));  // dz1 (s == 0 on pad columns)
    }
    if (vrow && hf == 0) *reinterpret_cast<f4*>(a.dz1 + (size_t)row * H2p + u * 16 + 4 * q) = acc1[u];
  }
  stamp();  // 8: exchange + layer-1 epilogue + output unit + dz1

  // ------------------------------------------------------------------ B0: dz0^T = W1 . dz1^T  (this wave's column half)
  __syncthreads();
  float* Wp = Wb + (size_t)H2p * H1p;
  chain_pf_store(pf, Wb, H2p, H1p, tid);
  if (a.kind == GOCTR_DIN) chain_pf_store(pf, Wp, H1p, Dp, tid, (H2p >> 2) * (H1p >> 2));
  __syncthreads();
  stamp();  // 9: W1T in LDS
  f4 dza[NT0H];
#pragma unroll
  for (int t = 0; t < NT0H; ++t) dza[t] = f4{0, 0, 0, 0};
  {
    const float* wp = Wb + ((size_t)q * H1p + t0 * 16 + i) * 4;
#pragma unroll
    for (int u = 0; u < NT1; ++u) {
      f4 w4[NT0H];
#pragma unroll
      for (int t = 0; t < NT0H; ++t) w4[t] = *reinterpret_cast<const f4*>(wp + t * 64);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < NT0H; ++t)
          if (t < NT0H - 1 || full) dza[t] = MF::mma(w4[t][r], acc1[u][r], dza[t]);
      wp += (size_t)4 * H1p * 4;
    }
  }
  stamp();  // 10: B0 MFMAs
#pragma unroll
  for (int t = 0; t < NT0H; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = (t0 + t) * 16 + 4 * q + r;
      const float k = DROP ? chain_dropk<DROP>(a.d0, a.st, row, n) : 1.0f;
      const float s = p0[t][r];
      dza[t][r] = (t < NT0H - 1 || full) ? (dza[t][r] * k) * (s * (1.0f - s)) : 0.0f;  // s == 0 on pad columns
    }
    if (vrow && (t < NT0H - 1 || full)) *reinterpret_cast<f4*>(a.dz0 + (size_t)row * H1p + (t0 + t) * 16 + 4 * q) = dza[t];
  }
  stamp();  // 11: dz0 epilogue
  if (a.kind != GOCTR_DIN) return;

  // ------------------------------------------------------------------ BP: dp^T = W0[U:U+D,:] . dz0^T (K split over the pair)
  const int ndp = Dp >> 4;
  f4 dpa[CHAIN_NDP];
#pragma unroll
  for (int v = 0; v < CHAIN_NDP; ++v) dpa[v] = f4{0, 0, 0, 0};
  {
    const float* wp = Wp + ((size_t)(t0 * 4 + q) * Dp + i) * 4;
#pragma unroll
    for (int t = 0; t < NT0H; ++t) {
      if (t < NT0H - 1 || full) {
#pragma unroll
        for (int v = 0; v < CHAIN_NDP; ++v) {
          if (v < ndp) {
            const f4 w4 = *reinterpret_cast<const f4*>(wp + v * 64);
#pragma unroll
            for (int r = 0; r < 4; ++r) dpa[v] = MF::mma(w4[r], dza[t][r], dpa[v]);
          }
        }
        wp += (size_t)4 * Dp * 4;
      }
    }
  }
  {
    float* xw = xch + ((rt * 2 + hf) * XS) * 64 + lane;
#pragma unroll
    for (int v = 0; v < CHAIN_NDP; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) xw[(v * 4 + r) * 64] = dpa[v][r];
    __syncthreads();
    const float* xr = xch + ((rt * 2 + (hf ^ 1)) * XS) * 64 + lane;
#pragma unroll
    for (int v = 0; v < CHAIN_NDP; ++v) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dpa[v][r] += xr[(v * 4 + r) * 64];
      if (vrow && hf == 0 && v < ndp) *reinterpret_cast<f4*>(a.dp + (size_t)row * Dp + v * 16 + 4 * q) = dpa[v];
    }
  }
  stamp();  // 12: dp done
}

}  // namespace goctr
