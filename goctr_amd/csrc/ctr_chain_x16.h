// ctr_chain_x16.h -- the training chain of ctr_chain_x3.h on 16-ROW tiles: two workgroups per CU (round 4 EXPERIMENT).
//
// RESULT: it LOST, and is therefore opt-in (GOCTR_CHAIN_X16=1 / 2 at model creation and launch; default: ctr_chain_x3_kernel).
// Parity is green (the 107 oracle tests of tests/test_gpu_{ctr,pipeline,fullsize,resume}.py pass with it as the default, and
// tests/test_gpu_ctr.py keeps one run of it), but at cfg3 the launch takes 25.8 us against 20.9 us (rocprofv3, same box): the
// per-wavefront stamps of a 16-row tile add up to the SAME ~37 k cycles as a 32-row tile's -- start-up 8.6 k, barriers and
// exchanges 7 k, the output unit 4 k are latencies that do not shrink with the row count -- and two co-resident workgroups share
// the MFMA pipe, the LDS port and the texture-address path instead of hiding each other's stalls.  Numbers and stamps:
// profiles/r04_chain_x16_ab.txt.  Kept for one round as the evidence VERDICT r3 item 3 asked for.
//
//   h0 -> sigma(h0.W0) -> drop -> sigma(.W1) -> drop -> sigma(.W2) -> BCE term, dz2 -> dz1 -> dz0 -> dp (-> attention backward)
//   (model/din/din.go:301-315, model/cost.go:9-17 and their hand-derived backward)
//
// Why.  ctr_chain_x3_kernel gives a workgroup 32 batch rows and needs 252 VGPRs x 8 wavefronts: ONE workgroup per CU.  At
// BASELINE configs[2] (B = 8192) that is exactly one tile per CU, and a tile is a five-stage DEPENDENT pipeline whose stages
// are bound by different units in turn (MFMA pipe in F0 / B0, the LDS port in the exchanges, VALU issue in the epilogues; DESIGN
// 4.2b): 15.8 us per tile however many follow on the same CU (profiles/r04_tile_sweep.txt), of which the MFMA pipe is busy
// 3.5 us.  Here a tile is 16 rows on v_mfma_f32_16x16x32_bf16 (C/D: col = lane & 15 = batch row, row = 4 (lane >> 4) + reg =
// feature), <= 128 VGPRs, ~61 KB of LDS: TWO workgroups per CU, four wavefronts per SIMD, whose stages drift apart -- one
// tile's epilogue / exchange runs under the other's MFMAs.  The price: every workgroup streams the whole weight images for half
// as many rows (247 MB instead of 120 MB per cfg3 launch out of L2), so the launcher uses this kernel only when the 32-row kernel
// would leave a CU with at most one tile (ceil(B / 32) <= CUs); larger batches keep ctr_chain_x3_kernel.
//
// Same arithmetic as ctr_chain_x3.h: every float32 value is the exact sum of three bf16 planes, a product is the 6-term
// split with float32 accumulation (small terms first, hi*hi last), products are computed transposed (Z^T = W^T . X^T) so that a
// wavefront's accumulators ARE the next product's B fragments: wavefront w owns the two 16-feature H1 tiles 2w, 2w + 1 (features
// 32w .. 32w + 31; 7 wavefronts cover H1p <= 224, the eighth helps with loads, the output unit and the attention backward), its
// 8 accumulator values per lane (tile tau, reg j -> feature 32w + 16 tau + 4q + j, q = lane >> 4) are the K = 32 chunk w of the
// layer-1 and dp products (K split, partial results meet in LDS).  K is chunked by 32 (the x3 kernel: by 16), so sums associate
// differently: the two kernels agree to float32 rounding, not bit for bit; both are bounded by the same oracle tests.
//
// Weight images (bf16 planes, kept by the Adam kernels / x3_build_images_kernel next to the x3 images; index functions below):
//   J0  W0   for F0:  [H1 tile t (14)][k chunk c (NK0)][plane][lane = 16q + m][8] = W0[32c + 8q + s][16t + m]
//   J1  W1   for F1:  [H1 chunk cc (7)][H2 tile u (5)][plane][lane = 16q + m][8]  = W1[perm16(cc, q, s)][16u + m]
//   J2  W1^T for B0:  [H1 tile t (14)][H2 chunk c (3)][plane][lane = 16q + m][8]  = W1[16t + m][32c + 8q + s]
//   J3  W0[U:U+32]^T for dp: [H1 chunk cc (7)][d tile v (2)][plane][lane = 16q + m][8] = W0[U + 16v + m][perm16(cc, q, s)]
// perm16(cc, q, s) = 32 cc + 16 (s / 4) + 4 q + s % 4: the H1 feature a lane holds at accumulator (tile s / 4, reg s % 4).
#pragma once
#include <hip/hip_runtime.h>

#include "ctr_chain_x3.h"

namespace goctr {

template <int NK0>
inline size_t chain_x16_lds_bytes() {
  // h0 fragment image | Z1 / dp exchange (7 partials x 5 tiles) | dz1 fragment image | z2 partials | dp / T rows
  return (size_t)NK0 * 3 * 1024 + (size_t)C16_NW * C16_NU * 1024 + (size_t)C16_NK2 * 3 * 1024 + 512 + 1024;
}

// 6-product bf16-split step for one 16x16x32 block, two accumulators (products that run over several K chunks)
#define C16_MMA6(AH, AC, A, B)                                                        \
  do {                                                                                \
    AC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2], B[0], AC, 0, 0, 0);            \
    AC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[2], AC, 0, 0, 0);            \
    AC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[1], AC, 0, 0, 0);            \
    AC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[0], AC, 0, 0, 0);            \
    AC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[1], AC, 0, 0, 0);            \
    AH = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[0], AH, 0, 0, 0);            \
  } while (0)
// the same for a product whose whole K is this one chunk (the wavefront's share of a K split): one accumulator, the small
// terms first, hi*hi last -- nothing is added to the sum afterwards, so a second accumulator would only be registers
#define C16_MMA6_ONE(ACC, A, B)                                                       \
  do {                                                                                \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2], B[0], ACC, 0, 0, 0);          \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[2], ACC, 0, 0, 0);          \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[1], ACC, 0, 0, 0);          \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[0], ACC, 0, 0, 0);          \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[1], ACC, 0, 0, 0);          \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[0], ACC, 0, 0, 0);          \
  } while (0)

// j-image pointers travel in ChainX3Args::j0 .. j3 (null: the model keeps no x16 images)
template <int NK0>
__global__ __launch_bounds__(512, 4) void ctr_chain_x16_kernel(ChainX3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c16_smem[];
  unsigned char* const h0img = c16_smem;                                                            // [NK0][3][64 lanes][16 B]
  float* const xch = reinterpret_cast<float*>(c16_smem + (size_t)NK0 * 3 * 1024);                   // [7 waves][5 tiles][64][4]
  unsigned char* const dz1img = reinterpret_cast<unsigned char*>(xch) + (size_t)C16_NW * C16_NU * 1024;   // [3][3][64][16 B]
  float* const z2p = reinterpret_cast<float*>(dz1img + (size_t)C16_NK2 * 3 * 1024);                 // [5][16]
  float* const dpl = z2p + 128;                                                                     // [16 rows][16] dp / T

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int tile = blockIdx.x;
  const int row = tile * 16 + n;
  const bool vrow = row < a.B;
  const int H1p = a.H1p, H2p = a.H2p, Ip = a.Ip;
  const bool own = w < C16_NW;                  // wavefront 7 owns no H1 tile: it multiplies wavefront 6's again, keep bits 0
  const int wt = own ? w : C16_NW - 1;
  const bool din = a.kind == GOCTR_DIN;

  unsigned long long ts[CX_NSTAMP];
#pragma unroll
  for (int k = 0; k < CX_NSTAMP; ++k) ts[k] = 0;
  auto stamp = [&](int k) { if (a.dbg) ts[k] = __builtin_amdgcn_s_memtime(); };
  stamp(0);

  // ---------------------------------------------------------------- h0: thread e < 64 NK0 owns entry (chunk e / 64, lane e % 64)
  // of the B-fragment image: h0[row e % 16][32 c + 8 (e % 64 / 16) .. + 8]
  cx_f4 hv[2] = {cx_f4{0.f, 0.f, 0.f, 0.f}, cx_f4{0.f, 0.f, 0.f, 0.f}};
  const int hc = tid >> 6, hl = tid & 63, hk = 32 * hc + 8 * (hl >> 4);
  const bool hent = tid < 64 * NK0;
  {
    const int hr = tile * 16 + (hl & 15);
    if (hent && hk < Ip && hr < a.B) {
      const float* hp = a.h0 + (size_t)hr * Ip + hk;
      hv[0] = *reinterpret_cast<const cx_f4*>(hp);
      hv[1] = *reinterpret_cast<const cx_f4*>(hp + 4);
    }
  }
  // ---------------------------------------------------------------- A-operand streams (global -> registers)
  const cx_u4* g0 = reinterpret_cast<const cx_u4*>(a.j0) + lane;        // + (((t*NK0 + c)*3 + p) * 64)
  const cx_u4* g1 = reinterpret_cast<const cx_u4*>(a.j1) + lane;        // + (((cc*NU + u)*3 + p) * 64)
  const cx_u4* g2 = reinterpret_cast<const cx_u4*>(a.j2) + lane;        // + (((t*NK2 + c)*3 + p) * 64)
  const cx_u4* g3 = reinterpret_cast<const cx_u4*>(a.j3) + lane;        // + (((cc*NV + v)*3 + p) * 64)
  constexpr int PF0 = NK0 < 3 ? NK0 : 3;
  cx_u4 ra0[PF0][2][3];
  auto load0 = [&](int c, int slot) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) ra0[slot][t][p] = *(g0 + (size_t)(((2 * wt + t) * NK0 + c) * 3 + p) * 64);
  };
#pragma unroll
  for (int c = 0; c < PF0; ++c) load0(c, c);

  const long long gr = a.st->batch_idx * (long long)a.B + row;
  const float y = (a.Y && vrow && gr < a.rows) ? a.Y[gr] : 0.f;

  // attention backward (see ChainX3Args::ab_*): this wavefront finishes samples 2w, 2w + 1 of the tile in attn_bwd_kernel<4,4,.>'s
  // lane layout (lane = 4 rl + dl: slot 16 p + rl, embedding columns 4 dl .. 4 dl + 3)
  const bool ab = din && a.ab_ids != nullptr;
  int abid[2] = {-1, -1};
  if (ab) {
    const long long b0 = a.st->batch_idx * (long long)a.B;
    const int lc = lane < a.ab_T ? lane : a.ab_T - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int b = tile * 16 + 2 * w + s;
      const int bc = b < a.B ? b : a.B - 1;
      const long long g = b0 + bc < a.rows ? b0 + bc : a.rows - 1;
      const int id = a.ab_ids[g * a.ab_T + lc];
      abid[s] = (b < a.B && lane < a.ab_T && b0 + b < a.rows && id >= 0 && id < a.ab_V) ? id : (int)a.ab_V;
    }
  }

  // zero half of the dz1 image's last chunk (H2 columns 80 .. 95: nobody writes them)
  if (tid < 96) {
    const int p = tid >> 5, l = 32 + (tid & 31);
    *reinterpret_cast<cx_u4*>(dz1img + ((size_t)((C16_NK2 - 1) * 3 + p) * 64 + l) * 16) = cx_u4{0u, 0u, 0u, 0u};
  }

  CxDrop dr0, dr1;
  dr0.init(a.d0, a.st, row, 32 * wt + 4 * q);
  dr1.init(a.d1, a.st, row, 0);

  // h0 -> bf16 planes, B-fragment image in LDS
  if (hent) {
    const float v[8] = {hv[0][0], hv[0][1], hv[0][2], hv[0][3], hv[1][0], hv[1][1], hv[1][2], hv[1][3]};
    cx_bf8 pl[3];
    cx_split8(v, pl);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<cx_bf8*>(h0img + ((size_t)(hc * 3 + p) * 64 + hl) * 16) = pl[p];
  }
  __syncthreads();                                            // (1) h0 image complete
  stamp(1);

  // ---------------------------------------------------------------- F0: Z0^T = W0^T . h0^T  (tiles 2 wt, 2 wt + 1)
  c16_acc ah0[2], ac0[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) { ah0[t] = c16_acc{0.f, 0.f, 0.f, 0.f}; ac0[t] = c16_acc{0.f, 0.f, 0.f, 0.f}; }
  // layer-1 columns this wavefront finishes after the exchange (wavefronts 0 .. 4): f = 16 w + 4 q + j
  const bool fin = w < C16_NU;
  const int fA = 16 * (fin ? w : 0) + 4 * q;
  // keep bits ride under the MFMAs: 8 of layer 0 (slot e = 4 tau + j: column 32 wt + 16 tau + 4 q + j), then 4 of layer 1
  auto draw_job = [&](int e) {
    if (e < 8) {
      const int cc = 16 * (e >> 2) + (e & 3);
      dr0.draw(cc, e, own && 32 * wt + 4 * q + cc < a.H1);
    } else if (e < 12) {
      const int f = fA + (e & 3);
      const bool k = (fin && f < a.H2) & ((mix32(dr1.hrow ^ ((uint32_t)f * 0x85EBCA6Bu + 0xC2B2AE35u)) >> 8) < dr1.thr);
      dr1.bits |= k ? (1u << (e - 8)) : 0u;
    }
  };
  constexpr int QDRAW = (12 + NK0 - 1) / NK0;
  cx_u4 ra1[C16_NU][3];                       // F1 A operands: [H2 tile u][plane] of this wavefront's chunk cc = wt
  auto load1 = [&](int u) {
#pragma unroll
    for (int p = 0; p < 3; ++p) ra1[u][p] = *(g1 + (size_t)((wt * C16_NU + u) * 3 + p) * 64);
  };
#pragma unroll
  for (int c = 0; c < NK0; ++c) {
    cx_bf8 bf[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const cx_bf8*>(h0img + ((size_t)(c * 3 + p) * 64 + lane) * 16);
    cx_bf8 af[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) af[t][p] = __builtin_bit_cast(cx_bf8, ra0[c % PF0][t][p]);
    // the A-operand stream does not stop at the product's end: a freed ring slot takes the next chunk, or two tiles of F1's operand
    if (c + PF0 < NK0) load0(c + PF0, c % PF0);
    else {
      const int k = c + PF0 - NK0;          // 0 .. PF0 - 1
      if (2 * k < C16_NU) load1(2 * k);
      if (2 * k + 1 < C16_NU) load1(2 * k + 1);
    }
    C16_MMA6(ah0[0], ac0[0], af[0], bf);
    C16_MMA6(ah0[1], ac0[1], af[1], bf);
#pragma unroll
    for (int e = c * QDRAW; e < (c + 1) * QDRAW && e < 12; ++e) draw_job(e);
    asm volatile("" : "+v"(dr0.bits), "+v"(dr1.bits));
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int u = 2 * PF0; u < C16_NU; ++u) load1(u);             // (tiles a short F0 had no slot for)
  stamp(2);

  // ---------------------------------------------------------------- layer-0 epilogue: sigmoid, dropout, A0 (registers + HBM)
  float p0[8], a0[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float s = chain_sigm(ah0[i >> 2][i & 3] + ac0[i >> 2][i & 3]);
    p0[i] = s;
    a0[i] = s * dr0.factor(i);
  }
  if (vrow && own) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int f = 32 * wt + 16 * t + 4 * q;
      if (f < H1p) *reinterpret_cast<cx_f4*>(a.A0 + (size_t)row * H1p + f) = cx_f4{a0[4 * t], a0[4 * t + 1], a0[4 * t + 2], a0[4 * t + 3]};
    }
  }
  stamp(3);

  // ---------------------------------------------------------------- F1: partial Z1^T = W1^T[:, own K] . A0^T[own K]
  c16_acc z1p[C16_NU];
  {
    cx_bf8 bf[3];
    cx_split8(a0, bf);
#pragma unroll
    for (int u = 0; u < C16_NU; ++u) {
      z1p[u] = c16_acc{0.f, 0.f, 0.f, 0.f};
      cx_bf8 af[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) af[p] = __builtin_bit_cast(cx_bf8, ra1[u][p]);
      C16_MMA6_ONE(z1p[u], af, bf);
    }
  }
  stamp(4);
  // B0 A operands in flight while the exchange and the output unit run: [tile][chunk][plane]
  cx_u4 ra2[2][C16_NK2][3];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < C16_NK2; ++c)
#pragma unroll
      for (int p = 0; p < 3; ++p) ra2[t][c][p] = *(g2 + (size_t)(((2 * wt + t) * C16_NK2 + c) * 3 + p) * 64);

  // ---------------------------------------------------------------- exchange 1: reduce-scatter of the partial Z1
  if (own) {
#pragma unroll
    for (int u = 0; u < C16_NU; ++u) *reinterpret_cast<c16_acc*>(xch + ((size_t)(w * C16_NU + u) * 64 + lane) * 4) = z1p[u];
  }
  __syncthreads();                                            // (2) partial Z1 visible
  stamp(5);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, w2v[4] = {0.f, 0.f, 0.f, 0.f};
  float part = 0.f;
  if (fin) {
    c16_acc z = c16_acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ws = 0; ws < C16_NW; ++ws)     // fixed wavefront order: bitwise reproducible
      z += *reinterpret_cast<const c16_acc*>(xch + ((size_t)(ws * C16_NU + w) * 64 + lane) * 4);
    const cx_f4 wv = *reinterpret_cast<const cx_f4*>(a.w2 + fA);       // (fA + 3 < H2p = 80)
    float a1v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = chain_sigm(z[r]);
      s1[r] = s;
      a1v[r] = s * dr1.factor(r);
      w2v[r] = wv[r];
      part += a1v[r] * wv[r];
    }
    if (vrow) *reinterpret_cast<cx_f4*>(a.A1 + (size_t)row * H2p + fA) = cx_f4{a1v[0], a1v[1], a1v[2], a1v[3]};
  }
  // ---------------------------------------------------------------- output unit: z2 = sum over the 5 x 4 partials of a row
  part += __shfl_xor(part, 16, 64);
  part += __shfl_xor(part, 32, 64);
  if (fin && q == 0) z2p[w * 16 + n] = part;
  __syncthreads();                                            // (3) z2 partials visible
  float z2 = z2p[n];
#pragma unroll
  for (int ws = 1; ws < C16_NU; ++ws) z2 += z2p[ws * 16 + n];
  const float yh = sigm_out(z2);
  const bool writer = w == 0 && q == 0 && vrow;
  const float one_eps = (float)(1.0 + 1e-8);
  const float dy = -((y / yh) - ((1.0f - y) / (one_eps - yh))) * a.inv_bglobal;
  const float d2 = dy * (yh * (1.0f - yh));
  if (writer) {
    a.yhat[row] = yh;
    a.lossrow[row] = logf(yh) * y + logf(one_eps - yh) * (1.0f - y);
    a.dz2[(size_t)row * 16] = d2;
  }
  // dz1 of the own features: HBM (for dW1 / dW2) and, as bf16 planes, the B-fragment image of the next product
  if (fin) {
    float dz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dz[r] = ((d2 * w2v[r]) * dr1.factor(r)) * (s1[r] * (1.0f - s1[r]));
    if (vrow) *reinterpret_cast<cx_f4*>(a.dz1 + (size_t)row * H2p + fA) = cx_f4{dz[0], dz[1], dz[2], dz[3]};
    unsigned int hh[2], mm[2], ll[2];
    tn_split3_pk(dz[0], dz[1], hh[0], mm[0], ll[0]);
    tn_split3_pk(dz[2], dz[3], hh[1], mm[1], ll[1]);
    // natural H2 order: f = 16 w + 4 q + j -> chunk c = f / 32, k group (f / 8) & 3, slots 4 (q & 1) .. + 3
    const int c = w >> 1, kq = 2 * (w & 1) + (q >> 1);
    unsigned char* d = dz1img + ((size_t)(c * 3) * 64 + kq * 16 + n) * 16 + (q & 1) * 8;
    *reinterpret_cast<cx_u2*>(d) = cx_u2{hh[0], hh[1]};
    *reinterpret_cast<cx_u2*>(d + 1024) = cx_u2{mm[0], mm[1]};
    *reinterpret_cast<cx_u2*>(d + 2048) = cx_u2{ll[0], ll[1]};
  }
  __syncthreads();                                            // (4) dz1 image complete
  stamp(6);

  // ---------------------------------------------------------------- B0: dz0^T = W1 . dz1^T  (tiles 2 wt, 2 wt + 1; K = H2)
  c16_acc ahb[2], acb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) { ahb[t] = c16_acc{0.f, 0.f, 0.f, 0.f}; acb[t] = c16_acc{0.f, 0.f, 0.f, 0.f}; }
  cx_u4 ra3[C16_NV][3];                        // dp A operands: [d tile][plane] of the own chunk
  const int nv = a.Dp > 16 ? 2 : 1;
#pragma unroll
  for (int c = 0; c < C16_NK2; ++c) {
    cx_bf8 bf[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const cx_bf8*>(dz1img + ((size_t)(c * 3 + p) * 64 + lane) * 16);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      cx_bf8 af[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) af[p] = __builtin_bit_cast(cx_bf8, ra2[t][c][p]);
      C16_MMA6(ahb[t], acb[t], af, bf);
    }
    if (c == 0 && din) {       // (the first chunk's operand registers are free: the dp operands take them)
#pragma unroll
      for (int v = 0; v < C16_NV; ++v)
#pragma unroll
        for (int p = 0; p < 3; ++p) ra3[v][p] = *(g3 + (size_t)((wt * C16_NV + (v < nv ? v : 0)) * 3 + p) * 64);
    }
  }
  // the behaviour rows of this wavefront's two samples, gates and similarity weights: in flight under the epilogue and the dp product
  float abx[2][4][4], abg[2] = {0.f, 0.f}, abw[2] = {0.f, 0.f};
  if (ab) {
    const int rl = lane >> 2, dl = lane & 3;
    const int lc = lane < a.ab_T ? lane : a.ab_T - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int id = __shfl(abid[s], 16 * p + rl, 64);
        const float4 t4 = *reinterpret_cast<const float4*>(a.ab_emb + 4 * dl + (size_t)(unsigned)id * 16);
        abx[s][p][0] = t4.x; abx[s][p][1] = t4.y; abx[s][p][2] = t4.z; abx[s][p][3] = t4.w;
      }
      const int b = tile * 16 + 2 * w + s;
      const int bc = b < a.B ? b : a.B - 1;
      const float g = a.ab_gate[(size_t)bc * a.ab_T + lc], wv = a.ab_wgt[(size_t)bc * a.ab_T + lc];
      const bool in = b < a.B && lane < a.ab_T;
      abg[s] = in ? g : 0.f;
      abw[s] = in ? wv : 0.f;
    }
  }
  stamp(7);
  float dzv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float s = p0[i];
    dzv[i] = ((ahb[i >> 2][i & 3] + acb[i >> 2][i & 3]) * dr0.factor(i)) * (s * (1.0f - s));     // factor == 0 on pad columns / wavefront 7
  }
  if (vrow && own) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int f = 32 * wt + 16 * t + 4 * q;
      if (f < H1p) *reinterpret_cast<cx_f4*>(a.dz0 + (size_t)row * H1p + f) = cx_f4{dzv[4 * t], dzv[4 * t + 1], dzv[4 * t + 2], dzv[4 * t + 3]};
    }
  }
  if (!din) {
    stamp(8);
    if (a.dbg && blockIdx.x == 0 && lane == 0) {
#pragma unroll
      for (int k = 0; k < CX_NSTAMP; ++k) a.dbg[w * CX_NSTAMP + k] = ts[k];
    }
    return;
  }

  // ---------------------------------------------------------------- BP: partial dp^T = W0[U:U+Dp, own K] . dz0^T[own K]
  c16_acc dpp[C16_NV];
  {
    cx_bf8 bf[3];
    cx_split8(dzv, bf);
#pragma unroll
    for (int v = 0; v < C16_NV; ++v) {
      dpp[v] = c16_acc{0.f, 0.f, 0.f, 0.f};
      if (v < nv) {
        cx_bf8 af[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) af[p] = __builtin_bit_cast(cx_bf8, ra3[v][p]);
        C16_MMA6_ONE(dpp[v], af, bf);
      }
    }
  }
  stamp(8);
  // exchange 2 (the Z1 area is free since barrier 3): d = 16 v + 4 q + j < Dp <= 32; wavefront v finishes tile v
  if (own) {
#pragma unroll
    for (int v = 0; v < C16_NV; ++v)
      if (v < nv) *reinterpret_cast<c16_acc*>(xch + ((size_t)(w * C16_NV + v) * 64 + lane) * 4) = dpp[v];
  }
  __syncthreads();                                            // (5) partial dp visible
  if (w < nv) {
    const int d0 = 16 * w + 4 * q;
    if (d0 < a.Dp) {
      c16_acc z = c16_acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ws = 0; ws < C16_NW; ++ws) z += *reinterpret_cast<const c16_acc*>(xch + ((size_t)(ws * C16_NV + w) * 64 + lane) * 4);
      if (vrow) *reinterpret_cast<cx_f4*>(a.dp + (size_t)row * a.Dp + d0) = cx_f4{z[0], z[1], z[2], z[3]};
      if (ab && w == 0) {
        const float Tf = (float)a.ab_T;        // dp / T once per element here (attn_bwd_kernel's first step)
        *reinterpret_cast<cx_f4*>(dpl + n * 16 + d0) = cx_f4{z[0] / Tf, z[1] / Tf, z[2] / Tf, z[3] / Tf};
      }
    }
  }
  if (ab) {
    __syncthreads();                                          // (6) the tile's dp rows visible
    const int dl = lane & 3, T = a.ab_T;
    const int src = (lane & 15) * 4, pw = lane >> 4;
    float term[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // same arithmetic, in the same order, as attn_bwd_kernel (ctr_kernels.h): the terms are bit-identical
      const cx_f4 dpt = *reinterpret_cast<const cx_f4*>(dpl + (2 * w + s) * 16 + 4 * dl);
      term[s] = 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float dg = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) dg += dpt[e] * abx[s][p][e];
        dg = group_sum<4>(dg);
        const float dgs = __shfl(dg, src, 64);
        if (pw == p) term[s] = dgs;
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int b = tile * 16 + 2 * w + s;
      if (b < a.B) {
        float* out = a.ab_out + (size_t)b * a.ab_Tp;
        if (lane < T) out[lane] = term[s] * (abg[s] * (1.0f - abg[s])) * abw[s];
        for (int t = T + lane; t < a.ab_Tp; t += 64) out[t] = 0.f;
      }
    }
  }
  stamp(9);
  if (a.dbg && blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int k = 0; k < CX_NSTAMP; ++k) a.dbg[w * CX_NSTAMP + k] = ts[k];
  }
}

}  // namespace goctr
