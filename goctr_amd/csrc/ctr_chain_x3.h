// ctr_chain_x3.h -- the fused per-row-tile chain of the DIN / YouTube TRAINING step on the 6-product bf16 split:
//
//   h0 -> sigma(h0.W0) -> drop -> sigma(.W1) -> drop -> sigma(.W2) -> BCE term, dz2 -> dz1 -> dz0 -> dp
//   (model/din/din.go:301-315, model/cost.go:9-17 and their hand-derived backward)
//
// Same function as ctr_chain.h (which stays for predict, explicit dropout masks and odd shapes); what changes is the
// arithmetic unit and, with it, the whole data flow:
//
// * Every float32 value x is the exact sum of three bf16 planes (x = hi + mid + lo, round-to-nearest splits) and a
//   product a*b is taken as hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid) on v_mfma_f32_32x32x16_bf16 with two
//   float32 accumulators (the small terms apart).  Measured against float64 this is MORE accurate than the f32-input
//   MFMA it replaces (tests bound it by the float32 oracle's own error), the six bf16 MFMAs cost 6 x 32
//   cycles per 32x32x16 block where v_mfma_f32_16x16x4_f32 costs 16 x 32, and -- unlike the f32-input MFMA on gfx950 --
//   they run UNDER the other wavefronts' loads and VALU work instead of blocking the SIMD.
// * Workgroup = 32 batch rows x 8 wavefronts (two per SIMD).  Products are computed TRANSPOSED, Z^T = W^T . X^T, so a
//   wavefront's accumulators hold Z[row = lane % 32][16 features of a 32-feature tile]: exactly the B-operand fragments
//   (k permuted, the weight images are laid out to match) of the NEXT product.  Wavefront w owns H1 tile w (layer 0 and
//   its backward: N split; 7 tiles, the eighth wavefront only helps with the elementwise work) and therefore a seventh
//   of the K range of the layer-1 / dp products (K split); the partial Z1 / dp meet in LDS (reduce-scatter).
// * Why two wavefronts per SIMD: a CDNA wavefront issues at most one instruction every four cycles, of whatever kind;
//   with one wavefront per SIMD (the first version of this kernel: 4 x 350 registers) every s_mov and ds_read costs
//   the same issue slot as a VALU or MFMA instruction and the kernel ran at instruction-count x 4 cycles (MFMA time
//   completely hidden, scripts/ubench/chain_x3_bench.hip).  Two wavefronts co-issue different instruction kinds.
// * With the rows of a workgroup in ONE wavefront's lanes no weight element is shared between wavefronts: the A
//   operands go straight from L2 into registers (global_load_dwordx4 of ready-made fragment images, 1 KiB per wave
//   instruction, three 16-k chunks ahead) -- no LDS staging, no loader wavefronts, no per-operand barriers.  All 256
//   workgroups stream the same 470 KB of images: L2 hits (scripts/ubench/l2_stream.hip: 55 B/clk/CU, 29 TB/s aggregate).
//
// Weight images (bf16 planes, written by the Adam kernel / x3_build_images_kernel, index functions in ctr_x3_images.h):
//   IMG0  W0   for F0:  [H1 tile t][k chunk c][plane][lane = 32 kg + m][8]   = W0[16c + 8kg + s][32t + m]
//   IMG1  W1   for F1:  [H1 chunk cc][H2 tile u][plane][lane = 32 kg + m][8] = W1[perm(cc, kg, s)][32u + m]
//   IMG2  W1^T for B0:  [H1 tile t][H2 chunk c][plane][lane = 32 kg + m][8]  = W1[32t + m][16c + 8kg + s]
//   IMG3  W0[U:U+D]^T for dp: [H1 chunk cc][plane][lane = 32 kg + d][8]      = W0[U + d][perm(cc, kg, s)]
// perm(cc, kg, s) = the H1 feature a lane holds at accumulator position: 32 (cc / 2) + 8 (2 (cc % 2) + s / 4) + 4 kg + s % 4.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/goctr.h"
#include "ctr_chain.h"
#include "ctr_kernels.h"
#include "mfma_gemm.h"

namespace goctr {

struct ChainX3Args {
  const float* h0; int Ip;                 // [B, Ip] float32 (attn_fwd)
  const unsigned short* img0; const unsigned short* img1; const unsigned short* img2; const unsigned short* img3;
  const float* w2;                         // the output unit's weight column, contiguous [H2p]
  int H1, H2, H1p, H2p, Dp, B; int kind;
  DropCfg d0, d1; const StepState* st;
  const float* Y; long long rows; float inv_bglobal;
  float* A0; float* A1; float* dz0; float* dz1; float* dz2; float* dp; float* yhat; float* lossrow;
  unsigned long long* dbg;
  // Attention backward inside this kernel (DIN, frozen embeddings, id mode, D == 16, T <= 64; ab_ids == null: the separate
  // attn_bwd_kernel does it).  The per-sample terms of the att0 gradient need dp -- which this kernel ends with -- and the
  // behaviour rows again; ids, gates and rows are requested while B0 / the dp product run, so the launch that used to
  // follow (6.0 us at cfg3: 410 k row gathers behind a kernel boundary) becomes ~50 instructions per sample at the tail.
  const int32_t* ab_ids; const float* ab_emb; long long ab_V; const float* ab_gate; const float* ab_wgt; const float* ab_fac; float* ab_out;   // ab_fac: (g (1 - g)) w ready-made (AttnArgs::fac), else gate and weight
  int ab_T, ab_Tp;
  // Round 6: gradients whose operands this launch holds in registers leave as PER-TILE sums instead (null: the operands are
  // stored and the weight-gradient launch multiplies them): tile_dw2 [tiles][H2p] = sum over the tile's rows of A1[row][f] *
  // dz2[row] (then A1 and dz2 are not stored at all), tile_att0 [tiles][ab_Tp] = the tile's sum of the att0 terms (then ab_out is
  // not stored).  The weight-gradient launch adds the tiles up (mfma_gemm.h tn_tile_sum_body).
  float* tile_dw2; float* tile_att0;
  int xcd_affine;     // training launches: workgroup -> tile by xcd_unit_of_block (GOCTR_XCD_AFFINE=0: workgroup b takes tile b)
};

constexpr int CX_NSTAMP = 16;

template <int NCH0>
inline size_t chain_x3_lds_bytes() {
  // h0 fragment image | Z1 / dp exchange (8 partials) | dz1 fragment image | z2 partials
  return (size_t)NCH0 * 3 * 1024 + (size_t)8 * CX_NU * 4 * 1024 + (size_t)CX_NCH2 * 3 * 1024 + 8 * 32 * 4;
}

// Experiment switches of scripts/ubench/chain_x3_bench.hip (never set in the library build): CX_EXP bit 0 = no A-operand
// loads after the first ring fill, bit 1 = no jobs under the MFMAs, bit 2 = no MFMAs.  Results are wrong, timings tell
// which resource bounds a phase.
#ifndef CX_EXP
#define CX_EXP 0
#endif

// 6-product bf16-split MFMA step for one 32x32x16 block: ah += hi*hi ; ac += everything else (smallest terms first)
#define CX_MMA6(AH, AC, A, B)                                                         \
  do {                                                                                \
    AC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[2], B[0], AC, 0, 0, 0);            \
    AC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[2], AC, 0, 0, 0);            \
    AC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[1], AC, 0, 0, 0);            \
    AC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1], B[0], AC, 0, 0, 0);            \
    AC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[1], AC, 0, 0, 0);            \
    AH = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0], B[0], AH, 0, 0, 0);            \
  } while (0)
#if CX_EXP & 4
#undef CX_MMA6
#define CX_MMA6(AH, AC, A, B) do { AC[0] += (float)A[0][0] + (float)B[1][1]; AH[1] += (float)A[2][0] + (float)B[0][1] + (float)A[1][3] + (float)B[2][2]; } while (0)
#endif

// 8 consecutive accumulator values -> the three planes of one B-operand fragment (slot s = position)
__device__ __forceinline__ void cx_split8(const float* v, cx_bf8 (&out)[3]) {
  unsigned int h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) tn_split3_pk(v[2 * j], v[2 * j + 1], h[j], m[j], l[j]);
  out[0] = __builtin_bit_cast(cx_bf8, cx_u4{h[0], h[1], h[2], h[3]});
  out[1] = __builtin_bit_cast(cx_bf8, cx_u4{m[0], m[1], m[2], m[3]});
  out[2] = __builtin_bit_cast(cx_bf8, cx_u4{l[0], l[1], l[2], l[3]});
}

// Keep bits of one layer as this lane sees them.  Bit `slot` set <=> the element is a real column (not padding) AND kept
// by the dropout hash (bit-identical to dropout_keep() / oracle orc_dropout_keep; without dropout: just "real column").
// The scale factor of an element is then  bit ? 1 / keep : 0  in the forward and in the backward pass.
//   * the three hash rounds that do not depend on the column are done once per lane (hrow); the column term
//     col * 0x85EBCA6B is (lane part) + (compile-time part): one add per element instead of a multiply;
//   * `u < keep` with u = (h >> 8) * 2^-24 is the integer comparison (h >> 8) < ceil(keep * 2^24) (both sides exact).
struct CxDrop {
  uint32_t hrow, thr, bits, cbase; float kv; bool on;
  // (gstep: the step counter as the kernel read it with its first state load -- read here, inside the tile loop, it was one more
  // load + wait + readfirstlane in front of the h0 split)
  __device__ __forceinline__ void init(const DropCfg& d, uint32_t gstep, int row, int col_lane) {
    on = d.mode == 2; bits = 0;
    const float keep = on ? 1.0f - d.p : 1.0f;
    kv = 1.0f / keep;
    uint32_t h = mix32(d.seed ^ 0x9E3779B9u);
    h = mix32(h ^ (gstep * 2u + d.layer));
    hrow = mix32(h ^ (d.row_off + (uint32_t)row));
    thr = on ? (uint32_t)ceilf(keep * 16777216.0f) : 0x1000000u;    // off: every 24-bit draw is below the threshold
    cbase = (uint32_t)col_lane * 0x85EBCA6Bu + 0xC2B2AE35u;
  }
  // element at column col_lane + col_const (col_const a compile-time constant), remembered as bit `slot`.  Branch-free
  // (the hash is drawn even when dropout is off: these instructions ride under MFMAs and a branch would split the
  // scheduling region they have to share with them)
  __device__ __forceinline__ void draw(int col_const, int slot, bool valid) {
    const bool k = valid & ((mix32(hrow ^ (cbase + (uint32_t)col_const * 0x85EBCA6Bu)) >> 8) < thr);
    bits |= k ? (1u << slot) : 0u;
  }
  __device__ __forceinline__ float factor(int slot) const { return (bits >> slot) & 1u ? kv : 0.0f; }
};


// Attention backward inside the chain kernel (see ChainX3Args::ab_*): wavefront w finishes samples 4 w .. 4 w + 3 of the
// tile, one at a time in attn_bwd_kernel<4, 4, .>'s lane layout (lane = 4 rl + dl: slot 16 p + rl, embedding columns
// 4 dl .. 4 dl + 3).  cx_ab_ids / cx_ab_gw: the slot ids / gates and similarity weights of its four samples, slot = lane;
// cx_ab_gather_one: one pass (16 behaviour rows) of one sample, 16 bytes per lane in flight under the products that follow.
template <class Args>
__device__ __forceinline__ void cx_ab_ids_load(const Args& a, int tile, int w, int lane, int (&idv)[4]) {
  const long long b0 = a.st->batch_idx * (long long)a.B;
  const int lc = lane < a.ab_T ? lane : a.ab_T - 1;
  // unconditional loads from clamped addresses, selects afterwards: a predicated load is a branch, and branches up here
  // split the scheduling region the first operand loads are issued from.  ALL FOUR loads first, then the tests (cx_ab_ids_check), and
  // the tests as bitwise ANDs: written as one short-circuit condition per sample, the compiler sank each load behind the tests in
  // front of it (b < B && slot < T && row < rows) and waited for it on the spot to evaluate the last two -- four dependent memory
  // round trips, each draining every operand load in flight, at the head of a launch whose first phase is a third of its duration.
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int b = tile * 32 + 4 * w + s;
    const int bc = b < a.B ? b : a.B - 1;
    const long long gr = b0 + bc < a.rows ? b0 + bc : a.rows - 1;
    idv[s] = a.ab_ids[gr * a.ab_T + lc];
  }
}
// (called behind the h0 split's barrier: the ids have the whole split to arrive in -- tested in the prologue, they were one more
// round trip in front of it)
template <class Args>
__device__ __forceinline__ void cx_ab_ids_check(const Args& a, int tile, int w, int lane, long long b0, const int (&idv)[4], int (&abid)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int b = tile * 32 + 4 * w + s;
    // missing / out-of-range ids, slots past T and rows past the dataset's end read the all-zero row V (like attn_bwd_kernel)
    const int ok = (int)(b < a.B) & (int)(lane < a.ab_T) & (int)(b0 + b < a.rows) & (int)(idv[s] >= 0) & (int)(idv[s] < a.ab_V);
    abid[s] = ok ? idv[s] : (int)a.ab_V;
  }
}
// (gates and similarity weights are only needed at the very end: requested late, 8 registers less through the products)
template <class Args>
__device__ __forceinline__ void cx_ab_gw(const Args& a, int tile, int w, int lane, float (&abf)[4]) {
  const int lc = lane < a.ab_T ? lane : a.ab_T - 1;
  // (all loads first, the tests as bitwise ANDs behind them -- see cx_ab_ids: "b < B && slot < T ? f : 0" per sample had become a
  // branch on the sample's (uniform) row test around load + full wait: four dependent round trips at the launch's tail, to lines the
  // previous launch wrote)
  if (a.ab_fac) {           // (uniform) the factor as the attention forward left it: one load per sample instead of two
    float fv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int b = tile * 32 + 4 * w + s;
      const int bc = b < a.B ? b : a.B - 1;
      fv[s] = a.ab_fac[(size_t)bc * a.ab_T + lc];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int b = tile * 32 + 4 * w + s;
      abf[s] = ((int)(b < a.B) & (int)(lane < a.ab_T)) ? fv[s] : 0.f;
    }
    return;
  }
  float gv[4], wv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int b = tile * 32 + 4 * w + s;
    const int bc = b < a.B ? b : a.B - 1;
    gv[s] = a.ab_gate[(size_t)bc * a.ab_T + lc]; wv[s] = a.ab_wgt[(size_t)bc * a.ab_T + lc];
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int b = tile * 32 + 4 * w + s;
    const float f = (gv[s] * (1.0f - gv[s])) * wv[s];       // (the same statement as AttnArgs::fac: the same bits)
    abf[s] = ((int)(b < a.B) & (int)(lane < a.ab_T)) ? f : 0.f;
  }
}
// Samples whose rows are requested inside F0's chunk loop, one or two gathers per chunk (the others: behind B0).  A gather
// instruction occupies the CU's texture-address path for ~40 cycles (16 separate 64-byte rows), and the 128 of them a
// tile needs stall their issuers when they come as one burst; under F0's MFMAs they are free, but each early sample holds
// 16 registers through the products.  Measured per step (cfg3, same box, three runs each): none early 50.5 us, 2 early 50.4,
// 3 early 49.8 (255 registers, none spilled); 4 early spills 40 registers and is 1 us slower than none.
// (Closing sessions of round 6: with the four id loads issued together (cx_ab_ids) three early samples no longer fit -- 256 registers
// and 8 spilled, the spill of a gathered row sitting behind a full s_waitcnt vmcnt(0) that drains the W0 ring in the middle of F0; two
// early samples are 244 registers, none spilled, and 41.6 against 42.0 us per step: profiles/r06_dependent_loads.txt.)
constexpr int CX_AB_EARLY = 2;
template <class Args>
__device__ __forceinline__ void cx_ab_gather_one(const Args& a, const int (&abid)[4], int lane, int s, int p, float (&abx)[4][4][4]) {
  const int rl = lane >> 2, dl = lane & 3;
  const int id = __shfl(abid[s], 16 * p + rl, 64);
  const float4 t4 = *reinterpret_cast<const float4*>(a.ab_emb + 4 * dl + (size_t)(unsigned)id * 16);
  abx[s][p][0] = t4.x; abx[s][p][1] = t4.y; abx[s][p][2] = t4.z; abx[s][p][3] = t4.w;
}

// (Round 6 built the attention FORWARD as this launch's head, as VERDICT r5 asked -- wavefront w gathering and pooling samples
// 4 w .. 4 w + 3 with attn_fwd_body's arithmetic, h0 rows / gates / weights handed on in LDS, the step's last launch a plain reduce +
// Adam: bit-identical to the separate launches, and 1.9 us per cfg3 step SLOWER (chain 20.8 -> 28.1 us, last launch 11.8 -> 6.0): the
// ~330 vector instructions per sample cost the same wherever they run, and here they sit in front of a dependent pipeline instead of
// beside a launch that waits on memory.  profiles/r06_chain_head.txt; commit 19008cd holds the code.)
// FWD: forward only (predict at launch sizes that give every CU a 32-row tile: the products and epilogues up to the output
// unit, no activations or deltas stored, no backward operands requested)
#ifndef CX_FWD_NFP
#define CX_FWD_NFP 4      // W0 chunks of the next tile requested a trip ahead by a forward-only launch (of the ring's 6)
#endif
template <int NCH0, bool FWD = false>
__global__ __launch_bounds__(512, 1) void ctr_chain_x3_kernel(ChainX3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cx_smem[];
  unsigned char* const h0img = cx_smem;                                         // [NCH0][3][64 lanes][16 B]
  float* const xch = reinterpret_cast<float*>(cx_smem + (size_t)NCH0 * 3 * 1024);   // [8 waves][NU][4 g][64][4]
  unsigned char* const dz1img = reinterpret_cast<unsigned char*>(xch) + (size_t)8 * CX_NU * 4 * 1024;   // [NCH2][3][64][16 B]
  float* const z2p = reinterpret_cast<float*>(dz1img + (size_t)CX_NCH2 * 3 * 1024);                     // [8][32]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int n = lane & 31, h = lane >> 5;             // (not const: laundered per trip of the forward-only tile loop, see there)
  // FWD launches are persistent over row tiles (tile, tile + gridDim.x, ...: see the loop below); a training launch has one
  // tile per workgroup
  const int ntiles = (a.B + 31) >> 5;
  // (FWD: the persistent workgroup's k-th tile is that of "workgroup" blockIdx + k * gridDim -- the same XCD when the grid is a
  // multiple of 8)
  int vblk = (int)blockIdx.x;
  int tile = a.xcd_affine ? xcd_unit_of_block(vblk, ntiles, 4) : vblk;
  const int tile_first = tile;
  int row = tile * 32 + n;
  bool vrow = row < a.B;
  int rowc = vrow ? row : a.B - 1;
  const int H1p = a.H1p, H2p = a.H2p, Ip = a.Ip;
  const bool own = w < CX_NT0;                 // wavefront 7 has no H1 tile: it multiplies tile 6 again, keep bits 0
  const int tt = own ? w : CX_NT0 - 1;
  const bool din = a.kind == GOCTR_DIN;

  unsigned long long ts[CX_NSTAMP];
#pragma unroll
  for (int k = 0; k < CX_NSTAMP; ++k) ts[k] = 0;
  auto stamp = [&](int k) { if (a.dbg) ts[k] = __builtin_amdgcn_s_memtime(); };
  stamp(0);
  if (FWD && a.dbg) { ts[14] = ts[0]; ts[15] = __builtin_amdgcn_s_memrealtime(); }   // (launch-wide: shader cycles against the 100 MHz clock)

  // ---------------------------------------------------------------- h0 rows of this wavefront's chunks: first loads out
  // chunk c is split by wavefront c % 8: lane (n, h) owns h0[row n][16c + 8h .. + 8]
  constexpr int NHQ = (NCH0 + 7) / 8;
  cx_f4 hv[NHQ][2];
  auto load_hv = [&](int rc) {
    const float* hp = a.h0 + (size_t)rc * Ip + 8 * h;
#pragma unroll
    for (int cq = 0; cq < NHQ; ++cq) {
      const int c = cq * 8 + w < NCH0 ? cq * 8 + w : NCH0 - 1;
      if (CX_EXP & 8) { hv[cq][0] = cx_f4{(float)c, 1.f, (float)lane, 2.f}; hv[cq][1] = hv[cq][0]; continue; }   // (experiment: no h0 loads)
      hv[cq][0] = *reinterpret_cast<const cx_f4*>(hp + c * 16);
      hv[cq][1] = *reinterpret_cast<const cx_f4*>(hp + c * 16 + 4);
    }
  };
  load_hv(rowc);
  // ---------------------------------------------------------------- A-operand streams (global -> registers)
  // (not const: the forward-only tile loop re-derives them per trip from a laundered lane index -- otherwise the ~60
  // loop-invariant 64-bit fragment addresses are hoisted out of the loop and spilled, 138 registers' worth)
  int lane_v = lane;
  const cx_u4* g0 = reinterpret_cast<const cx_u4*>(a.img0) + lane_v;        // + (((t*NCH0 + c)*3 + p) * 64)
  const cx_u4* g1 = reinterpret_cast<const cx_u4*>(a.img1) + lane_v;        // + (((cc*NU + u)*3 + p) * 64)
  const cx_u4* g2 = reinterpret_cast<const cx_u4*>(a.img2) + lane_v;        // + (((t*NCH2 + c)*3 + p) * 64)
  const cx_u4* g3 = reinterpret_cast<const cx_u4*>(a.img3) + lane_v;        // + ((cc*3 + p) * 64)

  // (ring depth: the forward-only tile loop carries the ring across its exchange / output-unit phases and has 12 registers less to spare)
  constexpr int CX_PF0 = NCH0 < 6 ? NCH0 : (FWD ? 5 : 6);
  cx_u4 ra0[CX_PF0][3];
  auto load0 = [&](int c, int slot) {
#pragma unroll
    for (int p = 0; p < 3; ++p) ra0[slot][p] = *(g0 + (size_t)((tt * NCH0 + c) * 3 + p) * 64);
  };
#pragma unroll
  for (int c = 0; c < CX_PF0; ++c) load0(c, c);

  const uint32_t gstep0 = a.st->gstep;          // (with the cursor below: ONE state load, ahead of everything that depends on it)
  float y = 0.f;
  if (!FWD) {
    const long long gr = a.st->batch_idx * (long long)a.B + row;
    y = (a.Y && vrow && gr < a.rows) ? a.Y[gr] : 0.f;
  }
  const bool ab = !FWD && din && a.ab_ids != nullptr;
  int abid[4] = {-1, -1, -1, -1}; float abf[4] = {0.f, 0.f, 0.f, 0.f};
  float abx[4][4][4];
  int abidv[4] = {0, 0, 0, 0};
  const long long ab_b0 = ab ? a.st->batch_idx * (long long)a.B : 0;
  if (ab) cx_ab_ids_load(a, tile, w, lane, abidv);
  // layer-1 columns this wavefront finishes after the exchange: group (u = w / 4, g = w % 4) and, for wavefronts 0..3,
  // (u = 2, g = w):  f = 32 u + 8 g + 4 h + r
  const int fA = 32 * (w >> 2) + 8 * (w & 3) + 4 * h, fB = 64 + 8 * (w & 3) + 4 * h;
  const bool hasB = w < 4;
  // (forward only: the output unit's weight columns of this wavefront, the same for every tile)
  cx_f4 w2pre[2] = {cx_f4{0.f, 0.f, 0.f, 0.f}, cx_f4{0.f, 0.f, 0.f, 0.f}};
  if (FWD) {
    if (fA < H2p) w2pre[0] = *reinterpret_cast<const cx_f4*>(a.w2 + fA);
    if (hasB && fB < H2p) w2pre[1] = *reinterpret_cast<const cx_f4*>(a.w2 + fB);
  }
  // Forward-only launches loop over their row tiles here.  A tile's start -- its h0 rows and the first six W0 chunks arriving,
  // 5.2 k of a tile's 18.3 k cycles (s_memtime of workgroup 0, GOCTR_DBG=chain) -- is requested while the previous tile's
  // exchange and output unit run (below, behind F1), so only a launch's first tile per CU pays it.  (Training: one trip.)
  for (;;) {
  if (FWD) {
    asm volatile("" : "+v"(lane_v), "+v"(n), "+v"(h));
    g0 = reinterpret_cast<const cx_u4*>(a.img0) + lane_v; g1 = reinterpret_cast<const cx_u4*>(a.img1) + lane_v;
    g2 = reinterpret_cast<const cx_u4*>(a.img2) + lane_v; g3 = reinterpret_cast<const cx_u4*>(a.img3) + lane_v;
  }
  if (FWD && tile != tile_first) {     // (a later trip: the ring slots its predecessor could not spare registers for)
    stamp(0);
#pragma unroll
    for (int c = CX_FWD_NFP; c < CX_PF0; ++c) load0(c, c);
  }
  CxDrop dr0, dr1;
  dr0.init(a.d0, gstep0, row, 32 * tt + 4 * h);
  dr1.init(a.d1, gstep0, row, 0);

  // h0 -> bf16 planes, B-fragment image in LDS
#pragma unroll
  for (int cq = 0; cq < NHQ; ++cq) {
    const int c = cq * 8 + w;
    if (c < NCH0) {
      float v[8] = {hv[cq][0][0], hv[cq][0][1], hv[cq][0][2], hv[cq][0][3], hv[cq][1][0], hv[cq][1][1], hv[cq][1][2], hv[cq][1][3]};
      if (!vrow) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
      }
      cx_bf8 pl[3];
      cx_split8(v, pl);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<cx_bf8*>(h0img + ((size_t)(c * 3 + p) * 64 + lane) * 16) = pl[p];
    }
  }
  __syncthreads();                                            // (1) h0 image complete
  stamp(1);
  if (ab) cx_ab_ids_check(a, tile, w, lane, ab_b0, abidv, abid);

  // ---------------------------------------------------------------- F0: Z0^T = W0^T . h0^T  (tile tt)
  cx_acc ah0, ac0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { ah0[i] = 0.f; ac0[i] = 0.f; }
  // keep bits ride under the MFMAs (they depend on nothing computed here): 16 of layer 0 (slot i: column
  // 32 tt + 8 (i / 4) + 4 h + i % 4), then 8 of layer 1 (slots 0..3: fA + r, slots 4..7: fB + r)
  auto draw_job = [&](int e) {
    if (FWD) {                       // no dropout at inference: the bits only mark the real columns, nothing is hashed
      if (e < 16) {
        const int cc = 8 * (e >> 2) + (e & 3);
        dr0.bits |= (own && 32 * tt + 4 * h + cc < a.H1) ? (1u << e) : 0u;
      } else if (e < 24) {
        const bool second = e >= 20;
        const int f = (second ? fB : fA) + (e & 3);
        dr1.bits |= (f < a.H2 && (!second || hasB)) ? (1u << (e - 16)) : 0u;
      }
      return;
    }
    if (e < 16) {
      const int cc = 8 * (e >> 2) + (e & 3);
      dr0.draw(cc, e, own && 32 * tt + 4 * h + cc < a.H1);
    } else if (e < 24) {
      const int r = e & 3;
      const bool second = e >= 20;
      const int f = (second ? fB : fA) + r;
      // (the column is not lane part + constant here: hash the full column)
      const bool k = (f < a.H2 && (!second || hasB)) &
                     ((mix32(dr1.hrow ^ ((uint32_t)f * 0x85EBCA6Bu + 0xC2B2AE35u)) >> 8) < dr1.thr);
      dr1.bits |= k ? (1u << (e - 16)) : 0u;
    }
  };
  constexpr int QDRAW = (24 + NCH0 - 1) / NCH0;
  // F1 A operands: [chunk j][u][plane] of this wavefront's two chunks cc = 2 tt + j.  The A-operand stream does not
  // stop at a product's end: once F0's ring stops refilling, its slots go to the first pieces of the next operand
  cx_u4 ra1[2][CX_NU][3];
  auto load1_piece = [&](int k) {      // k = 0 .. 5: (j, u) = (k / 3, k % 3)
    const int j = k / CX_NU, u = k - j * CX_NU;
#pragma unroll
    for (int p = 0; p < 3; ++p) ra1[j][u][p] = *(g1 + (size_t)(((2 * tt + j) * CX_NU + u) * 3 + p) * 64);
  };
#pragma unroll
  for (int c = 0; c < NCH0; ++c) {
    cx_bf8 bf[3], af[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const cx_bf8*>(h0img + ((size_t)(c * 3 + p) * 64 + lane) * 16);
#pragma unroll
    for (int p = 0; p < 3; ++p) af[p] = __builtin_bit_cast(cx_bf8, ra0[c % CX_PF0][p]);
    if (!(CX_EXP & 1) && c + CX_PF0 < NCH0) load0(c + CX_PF0, c % CX_PF0);
    else if (c + CX_PF0 - NCH0 < 2 * CX_NU) load1_piece(c + CX_PF0 - NCH0);
    CX_MMA6(ah0, ac0, af, bf);
    if (ab) {
      // a row gather or so per chunk, from the second chunk on (the ids are a dependent load behind the step state)
      constexpr int NE = 4 * CX_AB_EARLY;
      constexpr int C0 = NCH0 > 4 ? 1 : 0, PER = (NE + (NCH0 - C0) - 1) / (NCH0 - C0);
#pragma unroll
      for (int k = (c - C0) * PER; c >= C0 && k < (c - C0 + 1) * PER && k < NE; ++k) cx_ab_gather_one(a, abid, lane, k >> 2, k & 3, abx);
    }
    if (!(CX_EXP & 2)) {
#pragma unroll
      for (int e = c * QDRAW; e < (c + 1) * QDRAW && e < 24; ++e) draw_job(e);
    }
    // keep this chunk's draws with this chunk's MFMAs (instruction selection otherwise sinks them all behind the last
    // MFMA, where nothing hides them): the empty asm makes the bits so far an input of an ordered statement
    asm volatile("" : "+v"(dr0.bits), "+v"(dr1.bits));
    __builtin_amdgcn_sched_barrier(0);
  }
  stamp(2);

#pragma unroll
  for (int k = CX_PF0; k < 2 * CX_NU; ++k) load1_piece(k);     // (pieces a short F0 had no room for)

  // ---------------------------------------------------------------- layer-0 epilogue: sigmoid, dropout, A0 (registers + HBM)
  float p0[16];             // pre-dropout sigmoid (for the backward)
  float a0[16];             // post-dropout activation
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float s = chain_sigm(ah0[i] + ac0[i]);
    p0[i] = s;
    a0[i] = s * dr0.factor(i);
  }
  if (!FWD && vrow && own) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = 32 * tt + 8 * g + 4 * h;
      if (f < H1p) *reinterpret_cast<cx_f4*>(a.A0 + (size_t)row * H1p + f) = cx_f4{a0[4 * g], a0[4 * g + 1], a0[4 * g + 2], a0[4 * g + 3]};
    }
  }
  stamp(3);

  // ---------------------------------------------------------------- F1: partial Z1^T = W1^T[:, own K] . A0^T[own K]
  cx_acc ah1[CX_NU], ac1[CX_NU];
#pragma unroll
  for (int u = 0; u < CX_NU; ++u)
#pragma unroll
    for (int i = 0; i < 16; ++i) { ah1[u][i] = 0.f; ac1[u][i] = 0.f; }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    cx_bf8 bf[3];
    cx_split8(&a0[8 * j], bf);
#pragma unroll
    for (int u = 0; u < CX_NU; ++u) {
      cx_bf8 af[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) af[p] = __builtin_bit_cast(cx_bf8, ra1[j][u][p]);
      CX_MMA6(ah1[u], ac1[u], af, bf);
    }
  }
  stamp(4);
  // B0 A operands in flight while the exchanges run: [chunk c][plane]
  cx_u4 ra2[CX_NCH2][3];
  if (!FWD) {
#pragma unroll
    for (int c = 0; c < CX_NCH2; ++c)
#pragma unroll
      for (int p = 0; p < 3; ++p) ra2[c][p] = *(g2 + (size_t)((tt * CX_NCH2 + c) * 3 + p) * 64);
  }

  // ---------------------------------------------------------------- exchange 1: reduce-scatter of the partial Z1
  if (own) {
    float* xw = xch + ((size_t)(w * CX_NU) * 4 * 64 + lane) * 4;
#pragma unroll
    for (int u = 0; u < CX_NU; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<cx_f4*>(xw + (size_t)(u * 4 + g) * 256) =
            cx_f4{ah1[u][4 * g] + ac1[u][4 * g], ah1[u][4 * g + 1] + ac1[u][4 * g + 1], ah1[u][4 * g + 2] + ac1[u][4 * g + 2],
                  ah1[u][4 * g + 3] + ac1[u][4 * g + 3]};
  }
  int next_tile = tile;
  if (FWD) {
    // the next tile's rows and first weight chunks (workgroup-uniform branch): hv and the ring are free since F0, the layer-1
    // accumulators have just left for LDS.  In program order BEHIND every other load of this trip (the output unit's weights
    // are fetched before the loop): a wait for an earlier load never waits for these
    vblk += (int)gridDim.x;
    next_tile = vblk < ntiles ? (a.xcd_affine ? xcd_unit_of_block(vblk, ntiles, 4) : vblk) : ntiles;
    if (next_tile < ntiles) {
      const int nrow = next_tile * 32 + n;
      load_hv(nrow < a.B ? nrow : a.B - 1);
#pragma unroll
      for (int c = 0; c < CX_FWD_NFP && c < CX_PF0; ++c) load0(c, c);
    }
  }
  __syncthreads();                                            // (2) partial Z1 visible
  stamp(5);
  // this wavefront finishes group A = (u = w / 4, g = w % 4) and, wavefronts 0..3, group B = (u = 2, g = w)
  float s1[2][4], w2v[2][4];
  float part = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int u = k == 0 ? (w >> 2) : 2, g = w & 3;
    const int f0 = k == 0 ? fA : fB;
    const bool act = (k == 0 || hasB) && f0 < H2p;            // (wave-uniform except through h: f0 < H2p is per lane)
    cx_f4 z = cx_f4{0.f, 0.f, 0.f, 0.f};
    if (k == 0 || hasB) {
#pragma unroll
      for (int ws = 0; ws < CX_NT0; ++ws)   // fixed wavefront order: bitwise reproducible
        z += *reinterpret_cast<const cx_f4*>(xch + ((size_t)((ws * CX_NU + u) * 4 + g) * 64 + lane) * 4);
    }
    cx_f4 wv = cx_f4{0.f, 0.f, 0.f, 0.f};
    if (FWD) wv = w2pre[k];
    else if (act) wv = *reinterpret_cast<const cx_f4*>(a.w2 + f0);
    float a1v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = chain_sigm(z[r]);
      s1[k][r] = s;
      a1v[r] = s * dr1.factor(k * 4 + r);
      w2v[k][r] = wv[r];
      part += a1v[r] * wv[r];
    }
    if (!FWD && vrow && act && !a.tile_dw2) *reinterpret_cast<cx_f4*>(a.A1 + (size_t)row * H2p + f0) = cx_f4{a1v[0], a1v[1], a1v[2], a1v[3]};
  }
  // ---------------------------------------------------------------- output unit: z2 = sum over the 8 x 2 partials
  part += __shfl_xor(part, 32, 64);
  if (h == 0) z2p[w * 32 + n] = part;
  __syncthreads();                                            // (3) z2 partials visible
  float z2 = z2p[n];
#pragma unroll
  for (int ws = 1; ws < 8; ++ws) z2 += z2p[ws * 32 + n];
  const float yh = sigm_out(z2);
  const bool writer = w == 0 && h == 0 && vrow;
  if (FWD) {
    if (writer) a.yhat[row] = yh;
    stamp(6);
    if (a.dbg) ts[13] = __builtin_amdgcn_s_memrealtime();
    if (a.dbg && blockIdx.x == 0 && lane == 0) {      // (every trip: the last one's stamps survive -- a steady-state tile)
#pragma unroll
      for (int k = 0; k < CX_NSTAMP; ++k) a.dbg[w * CX_NSTAMP + k] = ts[k];
    }
    if (next_tile >= ntiles) return;
    tile = next_tile;
    row = tile * 32 + n; vrow = row < a.B; rowc = vrow ? row : a.B - 1;
    continue;
  }
  const float one_eps = (float)(1.0 + 1e-8);
  const float dy = -((y / yh) - ((1.0f - y) / (one_eps - yh))) * a.inv_bglobal;
  const float d2 = dy * (yh * (1.0f - yh));
  if (writer) {
    a.yhat[row] = yh;
    a.lossrow[row] = logf(yh) * y + logf(one_eps - yh) * (1.0f - y);
    if (!a.tile_dw2) a.dz2[(size_t)row * 16] = d2;
  }
  // dz1 of the own features: HBM (for dW1 / dW2) and, as bf16 planes, the B-fragment image of the next product
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int f0 = k == 0 ? fA : fB;
    const bool act = (k == 0 || hasB) && f0 < H2p;
    float dz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = s1[k][r];
      dz[r] = ((d2 * w2v[k][r]) * dr1.factor(k * 4 + r)) * (s * (1.0f - s));
    }
    if (a.tile_dw2 && (k == 0 || hasB)) {      // (wave-uniform)
      // dW2[f] of this tile: sum over its 32 rows (the lanes of this half) of A1[row][f] * dz2[row]; fixed lane order
      const float dv = vrow ? d2 : 0.f;
      cx_f4 pw;
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[r] = group_sum<32>((s1[k][r] * dr1.factor(k * 4 + r)) * dv);
      if (n == 0 && act) *reinterpret_cast<cx_f4*>(a.tile_dw2 + (size_t)tile * H2p + f0) = pw;
    }
    if (act) {
      if (vrow) *reinterpret_cast<cx_f4*>(a.dz1 + (size_t)row * H2p + f0) = cx_f4{dz[0], dz[1], dz[2], dz[3]};
      unsigned int hh[2], mm[2], ll[2];
      tn_split3_pk(dz[0], dz[1], hh[0], mm[0], ll[0]);
      tn_split3_pk(dz[2], dz[3], hh[1], mm[1], ll[1]);
      // natural H2 order: chunk c = f0 / 16, kg = (f0 / 8) & 1, slots 4h .. 4h + 3
      const int c = f0 >> 4, kg = (f0 >> 3) & 1;
      unsigned char* d = dz1img + ((size_t)(c * 3) * 64 + kg * 32 + n) * 16 + h * 8;
      *reinterpret_cast<cx_u2*>(d) = cx_u2{hh[0], hh[1]};
      *reinterpret_cast<cx_u2*>(d + 1024) = cx_u2{mm[0], mm[1]};
      *reinterpret_cast<cx_u2*>(d + 2048) = cx_u2{ll[0], ll[1]};
    }
  }
  // dp A operands (DIN): [chunk j][plane] of the own chunks
  cx_u4 ra3[2][3];
  if (din) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) ra3[j][p] = *(g3 + (size_t)((2 * tt + j) * 3 + p) * 64);
  }
  __syncthreads();                                            // (4) dz1 image complete
  stamp(6);

  // ---------------------------------------------------------------- B0: dz0^T = W1 . dz1^T  (tile tt, K = H2)
  cx_acc ahb, acb;
#pragma unroll
  for (int i = 0; i < 16; ++i) { ahb[i] = 0.f; acb[i] = 0.f; }
#pragma unroll
  for (int c = 0; c < CX_NCH2; ++c) {
    cx_bf8 bf[3], af[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const cx_bf8*>(dz1img + ((size_t)(c * 3 + p) * 64 + lane) * 16);
#pragma unroll
    for (int p = 0; p < 3; ++p) af[p] = __builtin_bit_cast(cx_bf8, ra2[c][p]);
    CX_MMA6(ahb, acb, af, bf);
  }
  if (ab) cx_ab_gw(a, tile, w, lane, abf);
  if (ab) {      // the behaviour rows of the remaining samples, in flight under the epilogue and the dp product
#pragma unroll
    for (int k = 4 * CX_AB_EARLY; k < 16; ++k) cx_ab_gather_one(a, abid, lane, k >> 2, k & 3, abx);
  }
  stamp(7);
  float dzv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float s = p0[i];
    dzv[i] = ((ahb[i] + acb[i]) * dr0.factor(i)) * (s * (1.0f - s));     // factor == 0 on pad columns / wavefront 7
  }
  if (vrow && own) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = 32 * tt + 8 * g + 4 * h;
      if (f < H1p) *reinterpret_cast<cx_f4*>(a.dz0 + (size_t)row * H1p + f) = cx_f4{dzv[4 * g], dzv[4 * g + 1], dzv[4 * g + 2], dzv[4 * g + 3]};
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

  // ---------------------------------------------------------------- BP: partial dp^T = W0[U:U+D, own K] . dz0^T[own K]
  cx_acc ahp, acp;
#pragma unroll
  for (int i = 0; i < 16; ++i) { ahp[i] = 0.f; acp[i] = 0.f; }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    cx_bf8 bf[3], af[3];
    cx_split8(&dzv[8 * j], bf);
#pragma unroll
    for (int p = 0; p < 3; ++p) af[p] = __builtin_bit_cast(cx_bf8, ra3[j][p]);
    CX_MMA6(ahp, acp, af, bf);
  }
  stamp(8);
  // exchange 2 (the Z1 area is free since barrier 3): d = 8g + 4h + r < Dp <= 32; wavefront g finishes group g
  if (own) {
    float* xw = xch + ((size_t)w * 4 * 64 + lane) * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<cx_f4*>(xw + (size_t)g * 256) = cx_f4{ahp[4 * g] + acp[4 * g], ahp[4 * g + 1] + acp[4 * g + 1],
                                                               ahp[4 * g + 2] + acp[4 * g + 2], ahp[4 * g + 3] + acp[4 * g + 3]};
  }
  __syncthreads();                                            // (5) partial dp visible
  if (w < 4) {
    const int d0 = 8 * w + 4 * h;
    if (d0 < a.Dp) {
      cx_f4 z = cx_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ws = 0; ws < CX_NT0; ++ws) z += *reinterpret_cast<const cx_f4*>(xch + ((size_t)(ws * 4 + w) * 64 + lane) * 4);
      if (vrow && !a.tile_att0) *reinterpret_cast<cx_f4*>(a.dp + (size_t)row * a.Dp + d0) = z;   // (tile_att0: its only reader is this launch's own tail)
      if (ab && d0 < 16) {
        // dp / T once per element here (attn_bwd_kernel's first step), not once per consumer lane
        const float Tf = (float)a.ab_T;
        *reinterpret_cast<cx_f4*>(reinterpret_cast<float*>(h0img) + n * 16 + d0) = cx_f4{z[0] / Tf, z[1] / Tf, z[2] / Tf, z[3] / Tf};   // (h0 image: free since F0)
      }
    }
  }
  if (ab) {
    __syncthreads();                                          // (6) the tile's dp rows visible
    const float* dpl = reinterpret_cast<const float*>(h0img);
    const int dl = lane & 3, T = a.ab_T;
    const int src = (lane & 15) * 4, pw = lane >> 4;
    // (the 16 (sample, pass) chains are independent and stay in one basic block: no early exit for rows past B, their
    // stores are predicated instead)
    float term[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // same arithmetic, in the same order, as attn_bwd_kernel (ctr_kernels.h): the terms are bit-identical
      const cx_f4 dpt = *reinterpret_cast<const cx_f4*>(dpl + (4 * w + s) * 16 + 4 * dl);
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
    if (a.tile_att0) {
      // the tile's att0 terms summed here: this wavefront's four samples in order, then the eight wavefronts in order
      float tsum = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int b = tile * 32 + 4 * w + s;
        tsum += (b < a.B && lane < T) ? term[s] * abf[s] : 0.f;
      }
      xch[w * 64 + lane] = tsum;                              // (the exchange area is free since barrier 6)
      __syncthreads();                                        // (7)
      if (w == 0 && lane < a.ab_Tp) {
        float sacc = xch[lane];
#pragma unroll
        for (int ws = 1; ws < 8; ++ws) sacc += xch[ws * 64 + lane];
        a.tile_att0[(size_t)tile * a.ab_Tp + lane] = sacc;
      }
    } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int b = tile * 32 + 4 * w + s;
      if (b < a.B) {
        float* out = a.ab_out + (size_t)b * a.ab_Tp;
        if (lane < T) out[lane] = term[s] * abf[s];
        for (int t = T + lane; t < a.ab_Tp; t += 64) out[t] = 0.f;
      }
    }
    }
  }
  stamp(9);
  if (a.dbg && blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int k = 0; k < CX_NSTAMP; ++k) a.dbg[w * CX_NSTAMP + k] = ts[k];
  }
  break;
  }   // (tile loop)
}

// ---------------------------------------------------------------- att0's update inside the weight-gradient launch
// The attention weights att0 (T floats) are the one parameter the step's LAST launch both updates and uses: reduce_attn_kernel's
// attention wavefronts -- the next batch's gather, gate and pooling -- wait ~2 us into their life for the flag of the reduce block that
// owns att0 (state -> slabs -> moments -> Adam -> write-through -> flag), and every one of the launch's 8192 wavefronts is resident and
// waiting by then: measured with the wait taken out (stale weights, same work) the step is 2.2 us shorter (profiles/r06_att0_early.txt).
// With the per-tile sums of the att0 terms out of the chain launch (ChainX3Args::tile_att0) their total needs no MFMA problem, only
// additions -- so ONE extra workgroup of the weight-gradient launch adds the tiles up and applies Adam to att0 right there: the new
// weights are in memory a launch boundary before anybody reads them, the last launch's attention part needs no flag and its reduce part
// leaves [offa, offa + Tp) alone (ReduceAdamArgs::skip_*).
// Same bits as the path it replaces, addition for addition: that path summed tiles [j tps, (j + 1) tps) in ascending order into slab j
// (mfma_gemm.h tn_tile_sum_body), then per float4 group lane pl of 8 added slabs [pl spp, (pl + 1) spp) in ascending order onto 0 and a
// xor-butterfly (1, 2, 4) added the eight lanes (ctr_kernels.h slab_group_sum), then adam_apply_pre with the step's corr1 / corr2.
struct Att0EarlyArgs {
  const float* tile_att0; int ntiles, Tp, tps, nslabs;     // [ntiles][Tp]; tiles per slab and slab count of the path this replaces
  AdamArgs ad; const StepState* st;
};
__device__ __forceinline__ void att0_early_body(const Att0EarlyArgs& e) {
  const int gid = (int)threadIdx.x;
  if (gid >= 2 * e.Tp) return;                               // 8 lanes per float4 group, Tp / 4 groups (whole wavefronts: Tp % 32 == 0)
  const int c0 = (gid >> 3) * 4, pl = gid & 7;               // columns c0 .. c0 + 3 of att0
  const int idx = e.ad.offa + c0 + pl;
  const bool mine = pl < 4 && c0 + pl < e.Tp;
  float w0 = 0.f, m0 = 0.f, v0 = 0.f;
  if (mine) { w0 = e.ad.W[idx]; m0 = e.ad.Mo[idx]; v0 = e.ad.Vo[idx]; }
  const float corr1 = e.st->corr1, corr2 = e.st->corr2;
  const int spp = (e.nslabs + 7) >> 3;
  const int lo = pl * spp;
  int hi = lo + spp; if (hi > e.nslabs) hi = e.nslabs;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = lo; j < hi; ++j) {
    const int t0 = j * e.tps;
    int t1 = t0 + e.tps; if (t1 > e.ntiles) t1 = e.ntiles;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int tb = t0; tb < t1; tb += 32) {                   // (32 tiles in flight: a slab of the cfg3 step in ONE round of loads)
      float4 v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const int t = tb + u < t1 ? tb + u : t1 - 1;
        v[u] = *reinterpret_cast<const float4*>(e.tile_att0 + (size_t)t * e.Tp + c0);
      }
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        const bool in = tb + u < t1;
        s.x += in ? v[u].x : 0.f; s.y += in ? v[u].y : 0.f; s.z += in ? v[u].z : 0.f; s.w += in ? v[u].w : 0.f;
      }
    }
    acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
    acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
  }
  if (mine) {
    const float g = pl == 0 ? acc.x : (pl == 1 ? acc.y : (pl == 2 ? acc.z : acc.w));
    adam_apply_pre(e.ad, idx, g, corr1, corr2, w0, m0, v0);
  }
}
// the wide weight-gradient launch with that workgroup behind its own (the last one: every workgroup of the launch has a CU to itself)
template <int KTW0, int KTW1>
__global__ __launch_bounds__(512) void gemm_tn_multi_x3w_att0_kernel(TnMulti a, Att0EarlyArgs e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char goctr_smem[];
  if (blockIdx.x == gridDim.x - 1) { att0_early_body(e); return; }
  tn_multi_x3w_block<KTW0, KTW1>(a, goctr_smem);
}

// (defined in ctr_fwd.hip)
int chain_x3_fwd_attributes();
// ctr_fwd4.h (also instantiated in ctr_fwd.hip): four wavefronts per tile, two workgroups per CU; Ip = 32, 144 or 240
int fwd4_attributes();
void launch_fwd4(int nch0, const ChainX3Args& a, dim3 grid, hipStream_t s);
void launch_chain_x3_fwd(int nch0, const ChainX3Args& a, dim3 grid, hipStream_t s);

}  // namespace goctr
