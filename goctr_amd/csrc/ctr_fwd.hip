// ctr_fwd.hip -- the forward-only instantiations of ctr_chain_x3_kernel (predict launches that give every CU a 32-row tile;
// model/model.go:214-352 PredictAbstract.Predict behind recommend.BatchPredict), in a translation unit of their own because
// they are compiled WITHOUT machine-level loop-invariant code motion (Makefile).
//
// The forward-only kernel is persistent over row tiles.  With machine LICM the nine float64 coefficient pairs of the output
// unit's exp() are hoisted out of the tile loop and then spilled; the scratch reloads queue, in order, behind the weight
// ring's outstanding loads, and the loop version of the kernel ran 25 % SLOWER than one workgroup per tile.  Switching the
// pass off for all of ctr.hip cost the weight-gradient kernel 0.5 us per step, so only this kernel lives here.
#define GOCTR_NO_PLAIN_KERNELS      // the headers' plain (non-template) kernels belong to ctr.hip
#include "common.h"
#include "ctr_chain_x3.h"
#include "ctr_fwd4.h"

namespace goctr {

// dynamic LDS above 64 KiB needs an explicit opt-in per kernel (goctr_init -> ctr.hip: set_kernel_attributes)
int chain_x3_fwd_attributes() {
  const void* const ks[3] = {reinterpret_cast<const void*>(ctr_chain_x3_kernel<2, true>),
                             reinterpret_cast<const void*>(ctr_chain_x3_kernel<9, true>),
                             reinterpret_cast<const void*>(ctr_chain_x3_kernel<15, true>)};
  for (const void* k : ks) GOCTR_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
  return 0;
}

void launch_chain_x3_fwd(int nch0, const ChainX3Args& a, dim3 grid, hipStream_t s) {
  switch (nch0) {
    case 2: hipLaunchKernelGGL((ctr_chain_x3_kernel<2, true>), grid, dim3(512), chain_x3_lds_bytes<2>(), s, a); break;
    case 9: hipLaunchKernelGGL((ctr_chain_x3_kernel<9, true>), grid, dim3(512), chain_x3_lds_bytes<9>(), s, a); break;
    default: hipLaunchKernelGGL((ctr_chain_x3_kernel<15, true>), grid, dim3(512), chain_x3_lds_bytes<15>(), s, a); break;
  }
}

int fwd4_attributes() {
  const void* const ks[3] = {reinterpret_cast<const void*>(ctr_fwd4_kernel<2, false>), reinterpret_cast<const void*>(ctr_fwd4_kernel<9, false>),
                             reinterpret_cast<const void*>(ctr_fwd4_kernel<15, true>)};
  for (const void* k : ks) GOCTR_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(80 * 1024)));
  return 0;
}

// (Ip = 144 with the exchange one H2 tile at a time measured the same as with one exchange: 626 / 627 / 614 against 631 / 612 / 608 M
// rows/s; Ip = 240 needs it to fit two workgroups per CU)
void launch_fwd4(int nch0, const ChainX3Args& a, dim3 grid, hipStream_t s) {
  if (nch0 == 2) hipLaunchKernelGGL((ctr_fwd4_kernel<2, false>), grid, dim3(256), (fwd4_lds_bytes<2, false>()), s, a);
  else if (nch0 == 9) hipLaunchKernelGGL((ctr_fwd4_kernel<9, false>), grid, dim3(256), (fwd4_lds_bytes<9, false>()), s, a);
  else hipLaunchKernelGGL((ctr_fwd4_kernel<15, true>), grid, dim3(256), (fwd4_lds_bytes<15, true>()), s, a);
}

}  // namespace goctr
