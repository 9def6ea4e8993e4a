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
//   F0   H1 tiles 2 w, 2 w + 1, ONE AFTER THE OTHER (wavefront 3 has no second tile: it multiplies tile 6 again and zeroes the
//        result -- its SIMD would wait at the next barrier anyway, and a conditional product costs the register allocation its
//        straight line): Z0^T tile = W0^T tile . h0^T, A operands (IMG0) streamed through a register ring, B operand = the
//        tile's h0 image in LDS.  The first tile's epilogue (16 sigmoids + the bf16 x 3 split of its activations, ~220 vector
//        instructions) rides under the second tile's MFMAs, the second tile's under the first two k chunks of F1's first H2 tile
//        (which need only the first tile's fragments): recommend +3 ... 5 % against both epilogues behind both products
//   F1   k chunks 4 w .. 4 w + 3 of H1 (its own activations, straight from the accumulators) x the 3 H2 tiles: partial Z1^T,
//        A operands (IMG1) as 12 pieces through a ring of F4_R1
//   exchange through LDS, wavefront w finishes columns 32 u + 8 w + 4 h + r (u = 0..2), then the output unit.
#pragma once
#include "ctr_chain_x3.h"

namespace goctr {

constexpr int F4_D0 = 2;       // W0 ring: 2 F4_D0 single-tile slots at Ip = 32, 3 at Ip >= 144 (4 spill 26 / 8 registers at Ip = 144 / 240)
constexpr int F4_R1 = 4;       // W1 pieces (one k chunk x one H2 tile) in flight per wavefront (3 measure the same, 6 spill 40 registers)

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

  // F0 runs its two H1 tiles one after the other (step sq = t NCH0 + c): the first tile's epilogue -- 16 sigmoids and the bf16 x 3
  // split of its activations, ~220 vector instructions -- rides under the second tile's MFMAs instead of running behind both with
  // the matrix cores idle.  One tile's fragments per ring slot.
  constexpr int NSQ = 2 * NCH0;
  constexpr int D0 = NSQ < 2 * F4_D0 ? NSQ : (NCH0 >= 9 ? 3 : 2 * F4_D0);     // (4 slots spill 26 / 8 registers at Ip = 144 / 240)
  cx_u4 ra0[D0][3];
  auto load0 = [&](int sq, int slot) {
    const int tt = sq < NCH0 ? t0 : t1, c = sq < NCH0 ? sq : sq - NCH0;
#pragma unroll
    for (int p = 0; p < 3; ++p) ra0[slot][p] = *(g0 + (size_t)((tt * NCH0 + c) * 3 + p) * 64);
  };
#pragma unroll
  for (int sq = 0; sq < D0; ++sq) load0(sq, sq);

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
    constexpr int QPC = (F4_R1 + D0 - 1) / D0;       // pieces requested per step of F0's tail
    cx_bf8 bfs[4][3];
    float a0[16];
    // the layer-0 epilogue of one H1 tile in six jobs: sigmoid (pad columns are zero) of four accumulator slots at a time, the
    // B fragments of a k chunk (cx_split8) after every second one
    auto epi_job = [&](int t, int j) {
      const int k = j / 3, r = j - 3 * k;              // k chunk 2 t + k: jobs 3 k, 3 k + 1 = its eight slots, 3 k + 2 = the split
      if (r < 2) {
#pragma unroll
        for (int i = 8 * k + 4 * r; i < 8 * k + 4 * r + 4; ++i) {
          const int col = 32 * (2 * w + t) + 8 * (i >> 2) + 4 * h + (i & 3);
          const float sg = chain_sigm(ah0[t][i] + ac0[t][i]);
          a0[i] = (col < a.H1 && (t == 0 || has1)) ? sg : 0.f;
        }
      } else cx_split8(&a0[8 * k], bfs[2 * t + k]);
    };
#pragma unroll
    for (int sq = 0; sq < NSQ; ++sq) {
      const int t = sq < NCH0 ? 0 : 1, c = sq < NCH0 ? sq : sq - NCH0;
      cx_bf8 bf[3], af[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const cx_bf8*>(h0img + ((size_t)(c * 3 + p) * 64 + lane) * 16);
#pragma unroll
      for (int p = 0; p < 3; ++p) af[p] = __builtin_bit_cast(cx_bf8, ra0[sq % D0][p]);
      if (sq + D0 < NSQ) load0(sq + D0, sq % D0);
      else {
#pragma unroll
        for (int q = (sq + D0 - NSQ) * QPC; q < (sq + D0 - NSQ + 1) * QPC && q < F4_R1; ++q) load1(q);
      }
      CX_MMA6(ah0[t], ac0[t], af, bf);
      if (t == 1) {
#pragma unroll
        for (int j = c * 6 / NCH0; j < (c + 1) * 6 / NCH0; ++j) epi_job(0, j);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp(2);
    // (the second tile's epilogue rides under the first two k chunks of F1's first H2 tile, which need only the first tile's fragments)

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
        for (int sq = 0; sq < D0; ++sq) load0(sq, sq);
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
        if (u == 0 && jj < 2) {
#pragma unroll
          for (int j = 3 * jj; j < 3 * jj + 3; ++j) epi_job(1, j);
        }
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
