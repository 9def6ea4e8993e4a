// ctr_serve.h -- a small serving pass (recommend.Rank / BatchPredict of a few hundred keys, recommend/api.go:106-131 ->
// rcmd.go:248-337) as ONE launch: key lookup + embedding gather + attention pooling, then the forward chain.
//
// The two-launch pass (attn_fwd_keys_kernel, ctr_fwd16_kernel) is two latency chains back to back: each sample's
// key -> behaviour window -> ids -> rows walk, then -- behind a kernel boundary -- 16 workgroups each waiting for their
// first 53 KB of W0 to arrive in LDS before the first MFMA.  Here a workgroup owns 16 rows end to end: 16 wavefronts, one
// sample each (attn_fwd_body in key mode, unchanged arithmetic: the scores are bit-identical), while the four loader
// wavefronts have already requested the first W0 block (LDS-DMA is fire-and-forget); the rows' h0 go through global
// memory to the four compute wavefronts (same CU: a wait for the stores + the workgroup barrier), wavefronts 8..15
// retire, and ctr_fwd16_body runs as before with its first operand already in LDS.
#pragma once
#include "ctr_chain.h"
#include "ctr_kernels.h"

namespace goctr {

// HV: see ctr_fwd16_body (10: Ip <= 160 -- no spill under this kernel's 128-register bound; 15: Ip <= 240, 16 spilled)
template <int LPR, int FAST, int HV>
__global__ __launch_bounds__(1024, 1) void ctr_serve16_kernel(AttnArgs aa, ChainArgs ca) {
  extern __shared__ __attribute__((aligned(16))) float chain_smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= 4 && wave < 8) {                    // the forward chain's loader wavefronts: first W0 block on its way
    ChainStager stg;
    const int kph0 = ca.Ip < CHAIN_KPH0 ? ca.Ip : CHAIN_KPH0;
    stg.begin(ca.W0i, chain_smem, kph0 * ca.H1p, wave - 4);
    stg.drain(lane);
  }
  // The workgroup's 16 keys with three coalesced loads by one wavefront (they sit in pinned HOST memory in a zero-copy pass:
  // 48 separate 4- and 8-byte reads per workgroup -- 768 for a 256-key pass -- queue on the PCIe read tags; 16.7 -> ?? us)
  // (in the forward chain's Z1 exchange area, untouched until the keys are long in registers: the kernel's LDS is all dynamic,
  // hipFuncAttributeMaxDynamicSharedMemorySize counts static LDS against the same 160 KB)
  AttnKey* const keys16 = reinterpret_cast<AttnKey*>(chain_smem + 2 * ca.buf_floats);
  if (wave == 0 && lane < 16) {
    const long long r = (long long)blockIdx.x * 16 + lane;
    AttnKey k{-1, -1, 0};
    if (r < aa.src.rows) { k.user = aa.src.k_users[r]; k.item = aa.src.k_items[r]; k.ts = aa.src.k_ts ? aa.src.k_ts[r] : 0; }
    keys16[lane] = k;
  }
  __syncthreads();
  const AttnKey mykey = keys16[wave];
  attn_fwd_body<4, LPR, FAST, true>(aa, 0, 0, aa.att0, nullptr, (int)blockIdx.x * 16 + wave, &mykey);
  // this wavefront's h0 row has left for L2 (the vector L1 is write-through and nobody in this launch read these lines
  // before); the barrier then orders it before the compute wavefronts' loads
  __builtin_amdgcn_s_waitcnt(0);                  // vmcnt(0) expcnt(0) lgkmcnt(0)
  __syncthreads();
  if (wave >= 8) return;                          // (retired wavefronts drop out of the later barriers)
  ctr_fwd16_body<4, 5, HV>(ca, true);
  // Wavefront 0 wrote the scores.  The failed flags left their wavefronts before the barrier above (s_waitcnt(0), then the
  // workgroup-scope barrier): both are ordered before the system-scope release below.
  if (ca.done && wave == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // (release only: no invalidate of the L2 under the workgroups still running)
    if (lane == 0) __hip_atomic_store(ca.done + blockIdx.x, ca.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace goctr
