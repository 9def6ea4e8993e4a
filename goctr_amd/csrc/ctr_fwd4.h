// ctr_fwd4.h -- forward-only chain for predict launches, four wavefronts per 32-row tile and TWO workgroups per CU
// (model/model.go:214-352 PredictAbstract.Predict behind recommend.BatchPredict; the layers are model/din.go:123-136 /
// model/youtube_dnn.go Fwd without dropout).
//
// Why a second forward-only kernel: ctr_chain_x3_kernel<., true> runs one 8-wavefront workgroup per CU (256 registers per
// lane leave room for no second one), so everything of a tile that is not a product -- the h0 split and its barrier, the
// exchange of the layer-1 partials, the second epilogue, the output unit behind two more barriers -- runs with the matrix
// cores idle: s_memtime of a steady-state tile (GOCTR_DBG=chain, profiles/r06_fwd4.txt) gives F0 3.4 k + F1 3.2 k cycles of
// products in a 13.4 k-cycle tile.  Here a tile belongs to FOUR wavefronts (one per SIMD) holding two H1 tiles each, the
// workgroup needs 75 KiB of LDS at Ip = 144 (77.5 KiB at Ip = 240 with the exchange one H2 tile at a time, template XU), and
// the CU holds two such workgroups that drift apart by themselves: while
// one sits in its exchange or output unit the other one's products have the matrix cores.  Same arithmetic as the
// 8-wavefront kernel up to the order in which the layer-1 partial sums are added (four partials of four k chunks instead of
// seven of two): float32 rounding, inside the 1e-5 parity bar like the 16-row kernel's (tests/test_gpu_ctr.py).
//
// Layout per wavefront w = 0..3, lane = 32 h + n (n = row of the tile):
//   F0   H1 tiles 2 w, 2 w + 1 (wavefront 3 has no second tile: it multiplies tile 6 again and zeroes the result -- its SIMD
//        would wait at the next barrier anyway, and a conditional product costs the register allocation its straight line):
//        Z0^T tile = W0^T tile . h0^T, A operands (IMG0) streamed through a register ring F4_D0 chunks deep, B operand =
//        the tile's h0 image in LDS, shared by both tiles
//   F1   k chunks 4 w .. 4 w + 3 of H1 (its own activations, straight from the accumulators) x the 3 H2 tiles: partial Z1^T,
//        A operands (IMG1) as 12 pieces through a ring of F4_R1
//   exchange through LDS, wavefront w finishes columns 32 u + 8 w + 4 h + r (u = 0..2), then the output unit.
#pragma once
#include "ctr_chain_x3.h"

namespace goctr {

// (3 chunks / 6 pieces in flight spill 26 / 40 registers; 3 pieces measure the same as 4)
constexpr int F4_D0 = 2;       // W0 chunks (of both tiles) in flight per wavefront
constexpr int F4_R1 = 4;       // W1 pieces (one k chunk x one H2 tile) in flight per wavefront

// XU: the exchange one H2 tile at a time through two 16 KiB halves (a barrier per tile instead of one for all three): 32 instead
// of 48 KiB, which is what lets Ip = 240 (45 KiB of h0 image) keep two workgroups per CU
template <int NCH0, bool XU>
inline size_t fwd4_lds_bytes() {
  // h0 fragment image | Z1 exchange (4 partials) | z2 partials
  return (size_t)NCH0 * 3 * 1024 + (size_t)4 * (XU ? 2 : CX_NU) * 4 * 1024 + 4 * 32 * 4;
}

template <int NCH0, bool XU>
__global__ __launch_bounds__(256, 2) void ctr_fwd4_kernel(ChainX3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char f4_smem[];
  unsigned char* const h0img = f4_smem;                                              // [NCH0][3][64 lanes][16 B]
  float* const xch = reinterpret_cast<float*>(f4_smem + (size_t)NCH0 * 3 * 1024);   // [4 waves][NU][4 g][64][4]
  float* const z2p = xch + (size_t)4 * (XU ? 2 : CX_NU) * 4 * 256;                   // [4][32]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int n = lane & 31, h = lane >> 5;             // (not const: laundered per trip of the tile loop)
  const int ntiles = (a.B + 31) >> 5;
  int vblk = (int)blockIdx.x;
  int tile = a.xcd_affine ? xcd_unit_of_block(vblk, ntiles, 4) : vblk;
  const int tile_first = tile;
  int row = tile * 32 + n;
  bool vrow = row < a.B;
  const int H2p = a.H2p, Ip = a.Ip;
  const bool has1 = 2 * w + 1 < CX_NT0;         // wavefront 3 holds one real H1 tile
  const int t0 = 2 * w, t1 = has1 ? 2 * w + 1 : 2 * w;

  unsigned long long ts[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ts[k] = 0;
  auto stamp = [&](int k) { if (a.dbg) ts[k] = __builtin_amdgcn_s_memtime(); };
  stamp(0);
  if (a.dbg) { ts[6] = ts[0]; ts[7] = __builtin_amdgcn_s_memrealtime(); }     // (launch-wide: shader cycles against the 100 MHz clock)

  // h0 rows: chunk c is split by wavefront c % 4, lane (n, h) owns h0[row n][16 c + 8 h .. + 8]
  constexpr int NHQ = (NCH0 + 3) / 4;
  cx_f4 hv[NHQ][2];
  auto load_hv = [&](int rc) {
    const float* hp = a.h0 + (size_t)rc * Ip + 8 * h;
#pragma unroll
    for (int cq = 0; cq < NHQ; ++cq) {
      const int c = cq * 4 + w < NCH0 ? cq * 4 + w : NCH0 - 1;
      hv[cq][0] = *reinterpret_cast<const cx_f4*>(hp + c * 16);
      hv[cq][1] = *reinterpret_cast<const cx_f4*>(hp + c * 16 + 4);
    }
  };
  load_hv(vrow ? row : a.B - 1);
  int lane_v = lane;
  const cx_u4* g0 = reinterpret_cast<const cx_u4*>(a.img0) + lane_v;        // + (((t*NCH0 + c)*3 + p) * 64)
  const cx_u4* g1 = reinterpret_cast<const cx_u4*>(a.img1) + lane_v;        // + (((cc*NU + u)*3 + p) * 64)

  constexpr int D0 = NCH0 < F4_D0 ? NCH0 : F4_D0;
  cx_u4 ra0[D0][2][3];
  auto load0 = [&](int c, int slot) {
#pragma unroll
    for (int p = 0; p < 3; ++p) ra0[slot][0][p] = *(g0 + (size_t)((t0 * NCH0 + c) * 3 + p) * 64);
#pragma unroll
    for (int p = 0; p < 3; ++p) ra0[slot][1][p] = *(g0 + (size_t)((t1 * NCH0 + c) * 3 + p) * 64);
  };
#pragma unroll
  for (int c = 0; c < D0; ++c) load0(c, c);

  // the output unit's weight columns of this wavefront: f = 32 u + 8 w + 4 h + r (the same for every tile)
  cx_f4 w2pre[CX_NU];
#pragma unroll
  for (int u = 0; u < CX_NU; ++u) {
    const int f0 = 32 * u + 8 * w + 4 * h;
    w2pre[u] = cx_f4{0.f, 0.f, 0.f, 0.f};
    if (f0 < H2p) w2pre[u] = *reinterpret_cast<const cx_f4*>(a.w2 + f0);
  }

  for (;;) {
    asm volatile("" : "+v"(lane_v), "+v"(n), "+v"(h));
    g0 = reinterpret_cast<const cx_u4*>(a.img0) + lane_v; g1 = reinterpret_cast<const cx_u4*>(a.img1) + lane_v;
    if (tile != tile_first) stamp(0);

    // h0 -> bf16 planes, B-fragment image in LDS
#pragma unroll
    for (int cq = 0; cq < NHQ; ++cq) {
      const int c = cq * 4 + w;
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

    // ---------------------------------------------------------------- F0: Z0^T = W0^T . h0^T  (tiles t0, t1)
    cx_acc ah0[2], ac0[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) { ah0[t][i] = 0.f; ac0[t][i] = 0.f; }
    // F1 A operands: piece q = 4 u + jj = (H2 tile u, k chunk 4 w + jj), slot q % F4_R1; the first F4_R1 pieces follow F0's
    // last ring refills (k chunks past the image's 14 -- wavefront 3's -- read chunk 13 again: their activations are zero)
    cx_u4 ra1[F4_R1][3];
    auto load1 = [&](int q) {
      const int u = q >> 2, jj = q & 3;
      const int cc = 4 * w + jj < CX_NCC ? 4 * w + jj : CX_NCC - 1;
#pragma unroll
      for (int p = 0; p < 3; ++p) ra1[q % F4_R1][p] = *(g1 + (size_t)((cc * CX_NU + u) * 3 + p) * 64);
    };
    constexpr int QPC = (F4_R1 + D0 - 1) / D0;       // pieces requested per chunk of F0's tail
#pragma unroll
    for (int c = 0; c < NCH0; ++c) {
      cx_bf8 bf[3], af[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const cx_bf8*>(h0img + ((size_t)(c * 3 + p) * 64 + lane) * 16);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[t][p] = __builtin_bit_cast(cx_bf8, ra0[c % D0][t][p]);
      if (c + D0 < NCH0) load0(c + D0, c % D0);
      else {
#pragma unroll
        for (int q = (c + D0 - NCH0) * QPC; q < (c + D0 - NCH0 + 1) * QPC && q < F4_R1; ++q) load1(q);
      }
      CX_MMA6(ah0[0], ac0[0], af[0], bf);
      CX_MMA6(ah0[1], ac0[1], af[1], bf);
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp(2);

    // ---------------------------------------------------------------- layer-0 epilogue: sigmoid (pad columns are zero), and
    // the activations as the B fragments of this wavefront's four k chunks (48 registers; the float32 values are not kept)
    cx_bf8 bfs[4][3];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float a0[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int col = 32 * (2 * w + t) + 8 * (i >> 2) + 4 * h + (i & 3);
        const float s = chain_sigm(ah0[t][i] + ac0[t][i]);
        a0[i] = (col < a.H1 && (t == 0 || has1)) ? s : 0.f;
      }
      cx_split8(&a0[0], bfs[2 * t]);
      cx_split8(&a0[8], bfs[2 * t + 1]);
    }

    // ---------------------------------------------------------------- F1: partial Z1^T = W1^T[:, own K] . A0^T[own K], one H2
    // tile at a time (32 accumulator registers live instead of 96), each handed to the exchange as soon as it is complete
    float* const xw = xch + ((size_t)(w * CX_NU) * 4 * 64 + lane) * 4;
    float part = 0.f;
    int next_tile = ntiles;
    // the next tile's rows and first weight chunks (workgroup-uniform branch), in program order behind every other load
    auto prefetch_next = [&]() {
      vblk += (int)gridDim.x;
      next_tile = vblk < ntiles ? (a.xcd_affine ? xcd_unit_of_block(vblk, ntiles, 4) : vblk) : ntiles;
      if (next_tile < ntiles) {
        const int nrow = next_tile * 32 + n;
        load_hv(nrow < a.B ? nrow : a.B - 1);
#pragma unroll
        for (int c = 0; c < D0; ++c) load0(c, c);
      }
    };
    // wavefront w finishes columns 32 u + 8 w + 4 h + r of H2 tile u from the four partials at `xb` ([ws][g = w][lane])
    auto finish = [&](int u, const float* xb, int ustride) {
      const int f0 = 32 * u + 8 * w + 4 * h;
      if (32 * u + 8 * w < H2p) {                               // (wave-uniform)
        cx_f4 z = cx_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ws = 0; ws < 4; ++ws)   // fixed wavefront order: bitwise reproducible
          z += *reinterpret_cast<const cx_f4*>(xb + ((size_t)(ws * ustride * 4 + w) * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s = chain_sigm(z[r]);
          part += (f0 + r < a.H2 ? s : 0.f) * w2pre[u][r];
        }
      }
    };
#pragma unroll
    for (int u = 0; u < CX_NU; ++u) {
      cx_acc ah1, ac1;
#pragma unroll
      for (int i = 0; i < 16; ++i) { ah1[i] = 0.f; ac1[i] = 0.f; }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int q = u * 4 + jj;
        cx_bf8 af[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) af[p] = __builtin_bit_cast(cx_bf8, ra1[q % F4_R1][p]);
        if (q + F4_R1 < 4 * CX_NU) load1(q + F4_R1);
        CX_MMA6(ah1, ac1, af, bfs[jj]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // this wavefront's partial of tile u: [w][u][g][lane], or (XU) half u % 2: [w][g][lane]
      float* const xu = XU ? xch + (size_t)(u & 1) * 4 * 4 * 256 + ((size_t)(w * 4) * 64 + lane) * 4 : xw + (size_t)(u * 4) * 256;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<cx_f4*>(xu + (size_t)g * 256) =
            cx_f4{ah1[4 * g] + ac1[4 * g], ah1[4 * g + 1] + ac1[4 * g + 1], ah1[4 * g + 2] + ac1[4 * g + 2], ah1[4 * g + 3] + ac1[4 * g + 3]};
      if (XU) {
        // a half is rewritten by tile u + 2: every wavefront has read tile u's partials before it arrives at the barrier of
        // tile u + 1, which the writers of tile u + 2 are behind
        if (u == CX_NU - 1) { stamp(3); prefetch_next(); }
        __syncthreads();                                        // (2.u) tile u's partials visible
        finish(u, xch + (size_t)(u & 1) * 4 * 4 * 256, 1);
      }
    }
    if (!XU) {
      stamp(3);
      prefetch_next();
      __syncthreads();                                          // (2) partial Z1 visible
    }
    stamp(4);
    if (!XU) {
#pragma unroll
      for (int u = 0; u < CX_NU; ++u) finish(u, xch + (size_t)(u * 4) * 256, CX_NU);
    }
    // ---------------------------------------------------------------- output unit: z2 = sum over the 4 x 2 partials
    part += __shfl_xor(part, 32, 64);
    if (h == 0) z2p[w * 32 + n] = part;
    __syncthreads();                                            // (3) z2 partials visible
    // (the output unit's float64 exp is ~500 cycles of one wavefront: a different one -- a different SIMD -- per trip and
    // per workgroup, +1.5 % rows/s against always wavefront 0)
    if (w == ((vblk / (int)gridDim.x + (int)blockIdx.x) & 3)) {
      float z2 = z2p[n];
#pragma unroll
      for (int ws = 1; ws < 4; ++ws) z2 += z2p[ws * 32 + n];
      const float yh = sigm_out(z2);
      if (h == 0 && vrow) a.yhat[row] = yh;
    }
    stamp(5);
    if (a.dbg && blockIdx.x == 0 && lane == 0) {                // (every trip: the last one's stamps survive)
#pragma unroll
      for (int k = 0; k < 8; ++k) a.dbg[w * CX_NSTAMP + k] = ts[k];
      a.dbg[w * CX_NSTAMP + 8] = __builtin_amdgcn_s_memrealtime();
    }
    if (next_tile >= ntiles) {
      if (a.dbg && tid == 0 && blockIdx.x < 960) {              // (per workgroup: start, end, where it ran)
        unsigned long long* d = a.dbg + 128 + 4 * (size_t)blockIdx.x;
        d[0] = ts[7]; d[1] = __builtin_amdgcn_s_memrealtime();
        d[2] = (unsigned long long)__builtin_amdgcn_s_getreg(63492);     // HW_REG_HW_ID
        d[3] = (unsigned long long)__builtin_amdgcn_s_getreg(63508);     // HW_REG_XCC_ID
      }
      return;
    }
    tile = next_tile;
    row = tile * 32 + n; vrow = row < a.B;
  }
}

}  // namespace goctr
