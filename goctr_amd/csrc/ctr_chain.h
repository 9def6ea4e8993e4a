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
// Weight operands live in HBM as ready-made LDS images [k/4][n][k%4] (kept in sync by the Adam kernel, one
// extra store per parameter): the 4 k-values one lane feeds to 4 consecutive MFMAs are one 16-byte
// ds_read_b128 (measured: one ds_read_b32 per MFMA costs 47 cycles/MFMA against 32 for register operands,
// scripts/ubench/mfma_rate.hip), and staging an operand is a straight copy done with the LDS-DMA path
// (global_load_lds_dwordx4: 1 KiB per wave instruction, no staging registers, no ds_write pass).  Rows of n
// are 16 B apart, so the 16 lanes of a q-group read one contiguous 256 B bank row: conflict-free.
//
// Two LDS weight buffers; four extra LOADER wavefronts (4..7, one per SIMD) issue the DMA of operand j+1
// right after the barrier that retires operand j-1, so the compute wavefronts' instruction streams carry
// nothing but ds_read_b128 + MFMA (+ epilogue VALU); inside a phase the ds_read_b128 of the next 16-k chunk
// are issued before the current chunk's MFMAs.  Measured on gfx950 (scripts/ubench/mfma_dma_overlap.hip):
// v_mfma_f32_* does NOT overlap with another wavefront's VALU, LDS or vector-memory work on the same SIMD
// (bf16 MFMA does) -- fp32 matrix time and DMA landing time add up, so the byte count of the staged
// operands (280 KB per workgroup) is part of the kernel's critical path, not hidden behind it.
//
// Workgroup = 4 compute wavefronts = 2 row tiles (32 batch rows) x 2 halves: the two wavefronts of a row tile split
// the H1 = 208 columns 7 + 6 tiles, so each holds half of A0 -- i.e. half of the K range of the next GEMM --
// and the partial Z1 (and later partial dp) are summed through a 20 KB LDS exchange.  8192 rows => 256
// workgroups => one per CU, 1024 wavefronts => one per SIMD.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/goctr.h"
#include "ctr_kernels.h"
#include "mfma_gemm.h"

namespace goctr {

constexpr int CHAIN_KPH0 = 80;   // rows of W0 per LDS phase (5 chunks of 16)
constexpr int CHAIN_NDP = 2;     // max 16-wide tiles of the pooled-embedding gradient (D <= 32)
constexpr int CHAIN_HV = 15;     // h0 fragments (16 columns each) a lane keeps in registers: Ip <= 240
constexpr int CHAIN_NSTAMP = 16;

struct ChainArgs {
  const float* h0; int Ip;                       // [B, Ip]
  // weight operands as LDS images [K/4][N][4]: W0 [Ip x H1p], W1 [H1p x H2p], W1^T [H2p x H1p],
  // W0[U:U+D,:]^T [H1p x Dp]; w2 = the output unit's weight column, contiguous [H2p]
  const float* W0i; const float* W1i; const float* W1Ti; const float* W0sTi; const float* w2;
  int H1, H2, H1p, H2p, Dp; int B;
  int train; int kind;
  DropCfg d0, d1; const StepState* st;
  const float* Y; long long rows; float inv_bglobal;
  int buf_floats;                                // size of each of the two LDS weight buffers (floats)
  // outputs (training only, except yhat)
  float* A0; float* A1; float* dz0; float* dz1; float* dz2; float* dp; float* yhat; float* lossrow;
  unsigned long long* dbg;                       // optional [CHAIN_NSTAMP] phase timestamps of block 0 / wave 0 (s_memtime)
  // ctr_serve16_kernel only: [workgroups] words in pinned host memory; a workgroup stores `epoch` behind a system-scope release
  // once its 16 scores are written -- the host watches them instead of waiting on the stream (null: it waits on the stream)
  unsigned* done; unsigned epoch;
};

// floats of one LDS weight buffer: the largest staged operand (+ one tile of slack when the second half of
// an odd tile count reads -- and discards -- one tile past the block)
inline int chain_buf_floats(int Ip, int H1p, int H2p) {
  const int kph = Ip < CHAIN_KPH0 ? Ip : CHAIN_KPH0;
  const int a = kph * H1p, b = H1p * H2p;
  return (a > b ? a : b) + (((H1p >> 4) & 1) ? 256 : 0);
}
template <int NT1>
inline size_t chain_lds_bytes(int Ip, int H1p, int H2p) {
  return sizeof(float) * ((size_t)2 * chain_buf_floats(Ip, H1p, H2p) + 4 * 64 * (NT1 * 4 > CHAIN_NDP * 4 ? NT1 * 4 : CHAIN_NDP * 4));
}

typedef float chain_f4 __attribute__((ext_vector_type(4)));

// Asynchronous copy of nfl floats (multiple of 4) of a weight image into LDS with the LDS-DMA path: every
// wave instruction moves 1 KiB (lane l -> dst + 16 l bytes); completion is covered by the vmcnt(0) of the
// next __syncthreads().  The copy is issued one instruction at a time (step) so that the issue cost hides
// in the shadow of the MFMAs of the phase it is interleaved with; drain() issues whatever is left.  The
// workgroups start at rotated positions so that the CUs of an XCD do not all hit the same L2 lines at once.
struct ChainStager {
  const float* src; float* dst; int nfl; int steps; int j; int rot; int wave;
  __device__ __forceinline__ void begin(const float* s, float* d, int n, int w) {
    src = s; dst = d; nfl = n; wave = w; j = 0;
    const int nchunks = (n + 255) >> 8;
    steps = (nchunks + 3) >> 2;
    rot = steps ? (int)(blockIdx.x % (unsigned)steps) : 0;
  }
  // every staged operand is a whole number of 16-k chunks = a multiple of 1 KiB (N % 16 == 0), so a step is
  // all-or-nothing for the wavefront: scalar bookkeeping + one DMA instruction, no vector ALU work that would
  // have to squeeze in between the MFMAs of the compute wavefront sharing the SIMD
  __device__ __forceinline__ void step(int lane) {
    if (j < steps) {
      int jj = j + rot;
      jj = jj >= steps ? jj - steps : jj;
      const int c = wave + 4 * jj;
      if (c * 256 < nfl) {
        const char* sbase = reinterpret_cast<const char*>(src + c * 256);
        const unsigned voff = (unsigned)lane * 16u;
        __builtin_amdgcn_global_load_lds(sbase + voff, (__attribute__((address_space(3))) void*)(dst + c * 256), 16, 0, 0);
      }
      ++j;
    }
  }
  __device__ __forceinline__ void drain(int lane) {
    while (j < steps) step(lane);
  }
};

// Dropout of one layer as this lane sees it.  The scale factor of element (row, col) is mask / keep (din.go:308);
// DROP: 0 none, 1 explicit mask, 2 counter hash (bit-identical to dropout_keep() / oracle orc_dropout_keep):
//   * the three hash rounds that do not depend on the column are done once per lane (hrow),
//   * `u < keep` with u = (h >> 8) * 2^-24 is the integer comparison (h >> 8) < ceil(keep * 2^24) (both sides exact),
//   * mask / keep is 0 or 1 / keep: one IEEE division per lane instead of one per element,
//   * the forward pass records the keep bits of its elements in a register bit mask that the backward pass reuses
//     (the hash ran twice per element before: 41 us instead of 20 us for the whole kernel at cfg3).
template <int DROP>
struct ChainDrop {
  uint32_t hrow, thr, bits; float kv, keep; bool on; const float* mrow;
  __device__ __forceinline__ void init(const DropCfg& d, const StepState* st, int row) {
    on = DROP != 0 && d.mode != 0;
    bits = 0; hrow = 0; thr = 0; kv = 1.0f; mrow = nullptr;
    keep = 1.0f - d.p;
    if (!on) return;
    kv = 1.0f / keep;
    if (DROP == 1) { mrow = d.mask + (size_t)row * d.mask_ld; return; }
    uint32_t h = mix32(d.seed ^ 0x9E3779B9u);
    h = mix32(h ^ (st->gstep * 2u + d.layer));
    hrow = mix32(h ^ (d.row_off + (uint32_t)row));
    thr = (uint32_t)ceilf(keep * 16777216.0f);
  }
  // forward: factor of column `col`, remembered as bit `slot`
  __device__ __forceinline__ float fwd(int col, int slot) {
    if (DROP == 0 || !on) return 1.0f;
    if (DROP == 1) return mrow[col] / keep;           // explicit masks (parity entry only): any float, exact m / keep
    const bool k = (mix32(hrow ^ ((uint32_t)col * 0x85EBCA6Bu + 0xC2B2AE35u)) >> 8) < thr;
    bits |= k ? (1u << slot) : 0u;
    return k ? kv : 0.0f;
  }
  __device__ __forceinline__ float bwd(int col, int slot) const {
    if (DROP == 0 || !on) return 1.0f;
    if (DROP == 1) return mrow[col] / keep;
    return (bits >> slot) & 1u ? kv : 0.0f;
  }
};

// branch-free variant of sigm_hidden
__device__ __forceinline__ float chain_sigm(float x) {
  float s = __builtin_amdgcn_rcpf(1.0f + __expf(-x));
  s = x > 15.f ? 1.0f : s;
  return x < -88.f ? 0.0f : s;
}

// One MFMA phase: acc[t] += sum over `nch` 16-k chunks of  Wimg(chunk)[tile t] x bfrag[chunk], the A operands
// read from LDS one chunk ahead of the MFMAs that use them.  wp = this lane's address of (chunk 0, tile 0),
// cstride = floats between chunks, tiles are 64 floats apart.  Always all NT tiles: the wave that owns one
// tile less multiplies a throw-away tile (it would otherwise wait for its partner at the next barrier anyway;
// a conditional MFMA costs accumulator shuffles in every group).
template <int NT, int NCH, typename BF>
__device__ __forceinline__ void chain_mma_phase(chain_f4 (&acc)[NT], const float* wp, int cstride, int nch, BF&& bfrag) {
  typedef chain_f4 f4;
  using MF = Mfma<float>;
  f4 wa[NT], wb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wa[t] = *reinterpret_cast<const f4*>(wp + t * 64);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c < nch) {
      f4 (&cur)[NT] = (c & 1) ? wb : wa;
      f4 (&nxt)[NT] = (c & 1) ? wa : wb;
      if (c + 1 < NCH && c + 1 < nch) {
#pragma unroll
        for (int t = 0; t < NT; ++t) nxt[t] = *reinterpret_cast<const f4*>(wp + (size_t)(c + 1) * cstride + t * 64);
      }
      const f4 b = bfrag(c);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = MF::mma(cur[t][r], b[r], acc[t]);
    }
  }
}

template <int NT0H, int NT1, int DROP>
__global__ __launch_bounds__(512, 1) void ctr_chain_kernel(ChainArgs a) {
  typedef chain_f4 f4;
  using MF = Mfma<float>;
  extern __shared__ __attribute__((aligned(16))) float chain_smem[];
  float* const bufP = chain_smem;
  float* const bufQ = chain_smem + a.buf_floats;
  float* const xch = chain_smem + 2 * a.buf_floats;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  // phase timestamps stay in scalar registers and are written once at the end: a store in flight would be
  // waited for by the vmcnt(0) of the next barrier and distort the very phases being timed
  unsigned long long ts[CHAIN_NSTAMP];
#pragma unroll
  for (int k = 0; k < CHAIN_NSTAMP; ++k) ts[k] = 0;
  auto stamp = [&](int k) { if (a.dbg) ts[k] = __builtin_amdgcn_s_memtime(); };
  auto flush_stamps = [&]() {
    if (a.dbg && blockIdx.x == 0 && tid == 0) {
#pragma unroll
      for (int k = 0; k < CHAIN_NSTAMP; ++k) a.dbg[k] = ts[k];
    }
  };
  stamp(0);

  const int kph0 = Ip < CHAIN_KPH0 ? Ip : CHAIN_KPH0;
  // ------------------------------------------------------------------ loader wavefronts (4..7)
  // Operand j goes to LDS buffer j & 1; it is issued right after the barrier that retires operand j-1's
  // phase... i.e. one full MFMA phase ahead of its use, and the vmcnt(0) of the loaders' next barrier makes
  // it visible.  The barrier sequence below mirrors the compute wavefronts' one for one.
  if (wave >= 4) {
    ChainStager stg;
    const int lw = wave - 4;
    stg.begin(a.W0i, bufP, kph0 * H1p, lw);
    stg.drain(lane);
    __syncthreads();                                   // B1: operand 0 landed
    int lpar = 0;
    for (int k0 = 0; k0 < Ip; k0 += CHAIN_KPH0) {
      float* oth = lpar ? bufP : bufQ;
      const int k1 = k0 + CHAIN_KPH0;
      if (k1 < Ip) stg.begin(a.W0i + (size_t)k1 * H1p, oth, (Ip - k1 < CHAIN_KPH0 ? Ip - k1 : CHAIN_KPH0) * H1p, lw);
      else stg.begin(a.W1i, oth, H1p * H2p, lw);
      stg.drain(lane);
      lpar ^= 1;
      __syncthreads();                                 // end of an F0 phase: next operand landed
    }
    // W1 sits in buffer lpar; W1^T goes to the other one while F1 multiplies
    stg.begin(a.W1Ti, lpar ? bufP : bufQ, a.train ? H2p * H1p : 0, lw);
    stg.drain(lane);
    __syncthreads();                                   // F1 exchange barrier
    if (a.train && a.kind == GOCTR_DIN) __syncthreads();  // dp exchange barrier
    return;
  }

  // ------------------------------------------------------------------ compute wavefronts (0..3)
  // every h0 fragment of this lane: one burst of 16-byte loads up front (host guarantees Ip <= 16*CHAIN_HV)
  const float* hp = a.h0 + (size_t)rowc * Ip + 4 * q;
  f4 hall[CHAIN_HV];
#pragma unroll
  for (int c = 0; c < CHAIN_HV; ++c)
    if (c * 16 < Ip) hall[c] = *reinterpret_cast<const f4*>(hp + c * 16);
  // small operands of the later phases, fetched while the first DMA flies
  f4 w2v[NT1];
#pragma unroll
  for (int u = 0; u < NT1; ++u) w2v[u] = *reinterpret_cast<const f4*>(a.w2 + u * 16 + 4 * q);
  float y = 0.f;
  if (a.train) {
    const long long gr = a.st->batch_idx * (long long)a.B + row;
    y = (a.Y && vrow && gr < a.rows) ? a.Y[gr] : 0.f;
  }

  // ------------------------------------------------------------------ F0: Z0^T = W0^T . h0^T
  f4 acc0[NT0H];
#pragma unroll
  for (int t = 0; t < NT0H; ++t) acc0[t] = f4{0, 0, 0, 0};
  int par = 0;   // LDS buffer holding the operand about to be consumed
  __syncthreads();
  stamp(1);  // first W0 phase landed
#pragma unroll
  for (int ph = 0; ph < CHAIN_HV * 16 / CHAIN_KPH0; ++ph) {
    const int k0 = ph * CHAIN_KPH0;
    if (k0 < Ip) {
      const int kph = Ip - k0 < CHAIN_KPH0 ? Ip - k0 : CHAIN_KPH0;
      float* cur = par ? bufQ : bufP;
      const float* wp = cur + ((size_t)q * H1p + t0 * 16 + i) * 4;
      chain_mma_phase<NT0H, CHAIN_KPH0 / 16>(acc0, wp, 16 * H1p, kph >> 4,
                                             [&](int c) { return hall[ph * (CHAIN_KPH0 / 16) + c < CHAIN_HV ? ph * (CHAIN_KPH0 / 16) + c : CHAIN_HV - 1]; });
      par ^= 1;
      stamp(10 + ph);  // MFMAs issued, before the barrier
      __syncthreads();
      stamp(2 + ph);  // phase done, next operand landed
    }
  }
  // here: W1 sits in buffer `par`, the other buffer is free
  float* const bufW1 = par ? bufQ : bufP;
  float* const bufB = par ? bufP : bufQ;
  // sigmoid + dropout in registers; acc0 becomes A0 (post-dropout), p0 keeps the pre-dropout sigmoid
  ChainDrop<DROP> dr0, dr1;
  dr0.init(a.d0, a.st, row);
  dr1.init(a.d1, a.st, row);
  f4 p0[NT0H];
#pragma unroll
  for (int t = 0; t < NT0H; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = (t0 + t) * 16 + 4 * q + r;
      float s = chain_sigm(acc0[t][r]);
      s = (n < a.H1 && (t < NT0H - 1 || full)) ? s : 0.0f;
      p0[t][r] = s;
      acc0[t][r] = DROP ? s * dr0.fwd(n, t * 4 + r) : s;
    }
    if (a.train && vrow && (t < NT0H - 1 || full))
      *reinterpret_cast<f4*>(a.A0 + (size_t)row * H1p + (t0 + t) * 16 + 4 * q) = acc0[t];
  }
  stamp(4);  // F0 epilogue

  // ------------------------------------------------------------------ F1: Z1^T = W1^T . A0^T (K split over the pair)
  f4 acc1[NT1];
#pragma unroll
  for (int u = 0; u < NT1; ++u) acc1[u] = f4{0, 0, 0, 0};
  {
    const float* wp = bufW1 + ((size_t)(t0 * 4 + q) * H2p + i) * 4;
    chain_mma_phase<NT1, NT0H>(acc1, wp, 16 * H2p, ntl, [&](int t) { return acc0[t]; });
  }
  stamp(5);  // F1 MFMAs
  constexpr int XS = NT1 * 4 > CHAIN_NDP * 4 ? NT1 * 4 : CHAIN_NDP * 4;
  {
    float* xw = xch + ((rt * 2 + hf) * XS) * 64 + lane;
#pragma unroll
    for (int u = 0; u < NT1; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) xw[(u * 4 + r) * 64] = acc1[u][r];
    stamp(12);
    __syncthreads();   // exchange visible; W1^T landed; W1 retired
    stamp(13);
    const float* xr = xch + ((rt * 2 + (hf ^ 1)) * XS) * 64 + lane;
#pragma unroll
    for (int u = 0; u < NT1; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc1[u][r] += xr[(u * 4 + r) * 64];  // a+b == b+a: both halves agree bitwise
  }
  // A operands of the dp product (this wave's K half of W0[U:U+D,:]^T) straight into registers
  f4 wdp[CHAIN_NDP][NT0H];
  const int ndp = Dp >> 4;
  if (a.train && a.kind == GOCTR_DIN) {
#pragma unroll
    for (int v = 0; v < CHAIN_NDP; ++v)
#pragma unroll
      for (int t = 0; t < NT0H; ++t)
        if (v < ndp && (t < NT0H - 1 || full))
          wdp[v][t] = *reinterpret_cast<const f4*>(a.W0sTi + ((size_t)((t0 + t) * 4 + q) * Dp + v * 16 + i) * 4);
  }
  f4 p1[NT1];
  float part = 0.f;
#pragma unroll
  for (int u = 0; u < NT1; ++u) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = u * 16 + 4 * q + r;
      float s = chain_sigm(acc1[u][r]);
      s = n < a.H2 ? s : 0.0f;
      p1[u][r] = s;
      const float post = DROP ? s * dr1.fwd(n, u * 4 + r) : s;
      acc1[u][r] = post;
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
  if (!a.train) { flush_stamps(); return; }
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
      const float k = DROP ? dr1.bwd(u * 16 + 4 * q + r, u * 4 + r) : 1.0f;
      const float s = p1[u][r];
      acc1[u][r] = ((d2 * w2v[u][r]) * k) * (s * (1.0f - s));  // dz1 (s == 0 on pad columns)
    }
    if (vrow && hf == 0) *reinterpret_cast<f4*>(a.dz1 + (size_t)row * H2p + u * 16 + 4 * q) = acc1[u];
  }
  stamp(6);  // exchange + layer-1 epilogue + output unit + dz1

  // ------------------------------------------------------------------ B0: dz0^T = W1 . dz1^T  (this wave's column half)
  f4 dza[NT0H];
#pragma unroll
  for (int t = 0; t < NT0H; ++t) dza[t] = f4{0, 0, 0, 0};
  {
    const float* wp = bufB + ((size_t)q * H1p + t0 * 16 + i) * 4;
    chain_mma_phase<NT0H, NT1>(dza, wp, 16 * H1p, NT1, [&](int u) { return acc1[u]; });
  }
  stamp(7);  // B0 MFMAs
#pragma unroll
  for (int t = 0; t < NT0H; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float k = DROP ? dr0.bwd((t0 + t) * 16 + 4 * q + r, t * 4 + r) : 1.0f;
      const float s = p0[t][r];
      dza[t][r] = (t < NT0H - 1 || full) ? (dza[t][r] * k) * (s * (1.0f - s)) : 0.0f;  // s == 0 on pad columns
    }
    if (vrow && (t < NT0H - 1 || full)) *reinterpret_cast<f4*>(a.dz0 + (size_t)row * H1p + (t0 + t) * 16 + 4 * q) = dza[t];
  }
  stamp(8);  // dz0 epilogue
  if (a.kind != GOCTR_DIN) { flush_stamps(); return; }

  // ------------------------------------------------------------------ BP: dp^T = W0[U:U+D,:] . dz0^T (K split over the pair)
  f4 dpa[CHAIN_NDP];
#pragma unroll
  for (int v = 0; v < CHAIN_NDP; ++v) dpa[v] = f4{0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < NT0H; ++t) {
    if (t < NT0H - 1 || full) {
#pragma unroll
      for (int v = 0; v < CHAIN_NDP; ++v) {
        if (v < ndp) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dpa[v] = MF::mma(wdp[v][t][r], dza[t][r], dpa[v]);
        }
      }
    }
  }
  {
    // second exchange area = the W1 buffer (retired by the barrier of the first exchange), so that a fast
    // wave cannot overwrite a slot its partner has not read yet
    float* xw = bufW1 + ((rt * 2 + hf) * XS) * 64 + lane;
#pragma unroll
    for (int v = 0; v < CHAIN_NDP; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) xw[(v * 4 + r) * 64] = dpa[v][r];
    __syncthreads();
    const float* xr = bufW1 + ((rt * 2 + (hf ^ 1)) * XS) * 64 + lane;
#pragma unroll
    for (int v = 0; v < CHAIN_NDP; ++v) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dpa[v][r] += xr[(v * 4 + r) * 64];
      if (vrow && hf == 0 && v < ndp) *reinterpret_cast<f4*>(a.dp + (size_t)row * Dp + v * 16 + 4 * q) = dpa[v];
    }
  }
  stamp(9);  // dp done
  flush_stamps();
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward-only variant for small batches (the recommend / predict path, PredBatchSize = 4096): 16 batch rows per
// workgroup instead of 32, so 4096 rows still put one workgroup on every CU, and the 13 H1 tiles are split over FOUR
// compute wavefronts (4 + 3 + 3 + 3): each wavefront issues 144 + 80 MFMAs instead of 252 + 140.  The four partial
// Z1 are exchanged through LDS and added in wavefront order by every wavefront.  Same operands, staging and loader
// wavefronts as ctr_chain_kernel; no dropout at inference (the reference's predict graph has p = 0, quirk Q1).
// (staged0: the loader wavefronts have already REQUESTED the first W0 block -- ctr_serve16_kernel, ctr_serve.h)
// HV: h0 fragments a lane keeps in registers (Ip <= 16 HV; 10 for Ip <= 160 leaves room under a 128-register bound)
template <int NT0Q, int NT1, int HV = CHAIN_HV>
__device__ __forceinline__ void ctr_fwd16_body(const ChainArgs& a, bool staged0) {
  typedef chain_f4 f4;
  extern __shared__ __attribute__((aligned(16))) float chain_smem[];
  float* const bufP = chain_smem;
  float* const bufQ = chain_smem + a.buf_floats;
  float* const xch = chain_smem + 2 * a.buf_floats;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, q = lane >> 4;
  const int H1p = a.H1p, H2p = a.H2p, Ip = a.Ip;
  const int kph0 = Ip < CHAIN_KPH0 ? Ip : CHAIN_KPH0;
  if (wave >= 4) {                         // loader wavefronts: the barrier sequence mirrors the compute wavefronts'
    ChainStager stg;
    const int lw = wave - 4;
    if (!staged0) {
      stg.begin(a.W0i, bufP, kph0 * H1p, lw);
      stg.drain(lane);
    }
    __syncthreads();
    int lpar = 0;
    for (int k0 = 0; k0 < Ip; k0 += CHAIN_KPH0) {
      float* oth = lpar ? bufP : bufQ;
      const int k1 = k0 + CHAIN_KPH0;
      if (k1 < Ip) stg.begin(a.W0i + (size_t)k1 * H1p, oth, (Ip - k1 < CHAIN_KPH0 ? Ip - k1 : CHAIN_KPH0) * H1p, lw);
      else stg.begin(a.W1i, oth, H1p * H2p, lw);
      stg.drain(lane);
      lpar ^= 1;
      __syncthreads();
    }
    __syncthreads();                       // Z1 exchange barrier
    return;
  }
  const int row = blockIdx.x * 16 + i;
  const bool vrow = row < a.B;
  const int rowc = vrow ? row : a.B - 1;
  const int NT0 = H1p >> 4;
  // tiles of this wavefront: 4,3,3,3 of 13 (or 4,4,3,3 of 14): wave w starts at min(w * NT0Q, ...) balanced from the front
  const int base = NT0 / 4, extra = NT0 - 4 * base;                   // `extra` wavefronts own base + 1 tiles
  const int t0 = wave * base + (wave < extra ? wave : extra);
  const int ntl = base + (wave < extra ? 1 : 0);
  const float* hp = a.h0 + (size_t)rowc * Ip + 4 * q;
  f4 hall[HV];
#pragma unroll
  for (int c = 0; c < HV; ++c)
    if (c * 16 < Ip) hall[c] = *reinterpret_cast<const f4*>(hp + c * 16);
  f4 w2v[NT1];
#pragma unroll
  for (int u = 0; u < NT1; ++u) w2v[u] = *reinterpret_cast<const f4*>(a.w2 + u * 16 + 4 * q);
  f4 acc0[NT0Q];
#pragma unroll
  for (int t = 0; t < NT0Q; ++t) acc0[t] = f4{0, 0, 0, 0};
  int par = 0;
  __syncthreads();
#pragma unroll
  for (int ph = 0; ph < HV * 16 / CHAIN_KPH0; ++ph) {
    const int k0 = ph * CHAIN_KPH0;
    if (k0 < Ip) {
      const int kph = Ip - k0 < CHAIN_KPH0 ? Ip - k0 : CHAIN_KPH0;
      float* cur = par ? bufQ : bufP;
      const float* wp = cur + ((size_t)q * H1p + t0 * 16 + i) * 4;    // (a wavefront with one tile less multiplies a throw-away one)
      chain_mma_phase<NT0Q, CHAIN_KPH0 / 16>(acc0, wp, 16 * H1p, kph >> 4,
                                             [&](int c) { return hall[ph * (CHAIN_KPH0 / 16) + c < HV ? ph * (CHAIN_KPH0 / 16) + c : HV - 1]; });
      par ^= 1;
      __syncthreads();
    }
  }
  float* const bufW1 = par ? bufQ : bufP;
#pragma unroll
  for (int t = 0; t < NT0Q; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = (t0 + t) * 16 + 4 * q + r;
      const float sg = chain_sigm(acc0[t][r]);
      acc0[t][r] = (n < a.H1 && t < ntl) ? sg : 0.0f;
    }
  f4 acc1[NT1];
#pragma unroll
  for (int u = 0; u < NT1; ++u) acc1[u] = f4{0, 0, 0, 0};
  {
    const float* wp = bufW1 + ((size_t)(t0 * 4 + q) * H2p + i) * 4;
    chain_mma_phase<NT1, NT0Q>(acc1, wp, 16 * H2p, ntl, [&](int t) { return acc0[t]; });
  }
  constexpr int XS = NT1 * 4;
  {
    float* xw = xch + (wave * XS) * 64 + lane;
#pragma unroll
    for (int u = 0; u < NT1; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) xw[(u * 4 + r) * 64] = acc1[u][r];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NT1; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float z = 0.f;                     // fixed wavefront order: every wavefront computes the same bits
#pragma unroll
        for (int w = 0; w < 4; ++w) z += xch[(w * XS + u * 4 + r) * 64 + lane];
        acc1[u][r] = z;
      }
  }
  float part = 0.f;
#pragma unroll
  for (int u = 0; u < NT1; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = u * 16 + 4 * q + r;
      const float sg = chain_sigm(acc1[u][r]);
      part += (n < a.H2 ? sg : 0.0f) * w2v[u][r];
    }
  float z2 = part + __shfl_xor(part, 16, 64);
  z2 += __shfl_xor(z2, 32, 64);
  if (wave == 0 && q == 0 && vrow) a.yhat[row] = sigm_out(z2);
}
template <int NT0Q, int NT1>
__global__ __launch_bounds__(512, 1) void ctr_fwd16_kernel(ChainArgs a) {
  ctr_fwd16_body<NT0Q, NT1>(a, false);
}

}  // namespace goctr
