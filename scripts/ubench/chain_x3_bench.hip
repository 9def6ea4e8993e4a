// Stand-alone timing of ctr_chain_x3_kernel<9> at cfg3 (B = 8192, Ip = 144, H1 = 200, H2 = 80, DIN) with random operands:
// which resource bounds each phase?  Build variants with -DCX_EXP=n (bit 0: no A loads after the first ring fill, bit 1:
// no jobs under the MFMAs, bit 2: no MFMAs); prints the kernel's average duration and the s_memtime phase stamps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../goctr_amd/csrc/ctr_chain_x3.h"
using namespace goctr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const int B = 8192, Ip = 144, H1 = 200, H2 = 80, H1p = 208, H2p = 80, Dp = 16, NCH0 = 9;
  float *h0, *w2, *Y, *A0, *A1, *dz0, *dz1, *dz2, *dp, *yhat, *loss; unsigned short* img; StepState* st; unsigned long long* dbg;
  const size_t ni = cx_images_elems(NCH0);
  CK(hipMalloc(&h0, (size_t)B * Ip * 4)); CK(hipMalloc(&w2, 96 * 4)); CK(hipMalloc(&Y, B * 4));
  CK(hipMalloc(&A0, (size_t)B * H1p * 4)); CK(hipMalloc(&A1, (size_t)B * H2p * 4)); CK(hipMalloc(&dz0, (size_t)B * H1p * 4));
  CK(hipMalloc(&dz1, (size_t)B * H2p * 4)); CK(hipMalloc(&dz2, (size_t)B * 16 * 4)); CK(hipMalloc(&dp, (size_t)B * Dp * 4));
  CK(hipMalloc(&yhat, B * 4)); CK(hipMalloc(&loss, B * 4)); CK(hipMalloc(&img, ni * 2)); CK(hipMalloc(&st, sizeof(StepState)));
  CK(hipMalloc(&dbg, 8 * 16 * 8));
  std::vector<float> hh((size_t)B * Ip); for (auto& v : hh) v = (float)rand() / RAND_MAX;
  std::vector<unsigned short> hi(ni); for (auto& v : hi) v = (unsigned short)(0x3C00 + (rand() & 0x1FF)) ^ (rand() & 1 ? 0x8000 : 0);   // ~ +-0.01
  std::vector<float> hw(96, 0.1f), hy(B, 1.0f);
  StepState s0{0, 0, 0, 1};
  CK(hipMemcpy(h0, hh.data(), hh.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(img, hi.data(), ni * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w2, hw.data(), 96 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Y, hy.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(st, &s0, sizeof s0, hipMemcpyHostToDevice));
  ChainX3Args a{};
  a.h0 = h0; a.Ip = Ip; a.img0 = img; a.img1 = img + cx_img0_elems(NCH0); a.img2 = a.img1 + cx_img1_elems(); a.img3 = a.img2 + cx_img2_elems();
  a.w2 = w2; a.H1 = H1; a.H2 = H2; a.H1p = H1p; a.H2p = H2p; a.Dp = Dp; a.B = B; a.kind = GOCTR_DIN;
  a.d0 = DropCfg{2, 0.005f, nullptr, H1, 42u, 0u, 0u}; a.d1 = DropCfg{2, 0.005f, nullptr, H2, 42u, 1u, 0u};
  a.st = st; a.Y = Y; a.rows = B; a.inv_bglobal = 1.0f / B;
  a.A0 = A0; a.A1 = A1; a.dz0 = dz0; a.dz1 = dz1; a.dz2 = dz2; a.dp = dp; a.yhat = yhat; a.lossrow = loss; a.dbg = nullptr;
  CK(hipFuncSetAttribute((const void*)ctr_chain_x3_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const size_t lds = chain_x3_lds_bytes<9>();
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(ctr_chain_x3_kernel<9>, dim3(B / 32), dim3(512), lds, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int N = 100;
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(ctr_chain_x3_kernel<9>, dim3(B / 32), dim3(512), lds, 0, a);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  a.dbg = dbg;
  hipLaunchKernelGGL(ctr_chain_x3_kernel<9>, dim3(B / 32), dim3(512), lds, 0, a);
  CK(hipDeviceSynchronize());
  unsigned long long h[8][16]; CK(hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost));
  printf("CX_EXP=%d: %.2f us/launch (back to back); per wavefront, cycles since the workgroup's first stamp at: b1(h0 image) F0 epi0 F1 b2(Z1 xchg) b4(dz1 image) B0 BP end\n", CX_EXP, ms * 1e3 / N);
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < 8; ++w) if (h[w][0] < t0) t0 = h[w][0];
  const int ks[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
  for (int w = 0; w < 8; ++w) {
    printf("  w%d:", w);
    for (int k : ks) printf(" %6lld", (long long)(h[w][k] - t0));
    printf("\n");
  }
  return 0;
}
