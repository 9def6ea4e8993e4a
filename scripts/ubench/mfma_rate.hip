// calibrate v_mfma_f32_16x16x4_f32 / 32x32x2 issue rates, with and without one ds_read per MFMA
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* cyc, int iters) {
  __shared__ float sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = (float)(i & 7) * 0.001f;
  __syncthreads();
  f4 acc[NACC];
  for (int t = 0; t < NACC; ++t) acc[t] = f4{0,0,0,0};
  float b = threadIdx.x * 0.001f;
  const float* p = sm + (threadIdx.x & 63);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        float a = LDS ? p[(r * 212 + t * 16) & 8191] : b;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
      }
    }
    if (LDS) p += 4 * 212; if (LDS && p > sm + 4000) p -= 3392;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, unsigned long long* cyc, int iters) {
  f16 acc[NACC];
  for (int t = 0; t < NACC; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0;
  float b = threadIdx.x * 0.001f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[t], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][5];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* d; unsigned long long* c; hipMalloc(&d, 1 << 24); hipMalloc(&c, 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](const char* name, auto launch, double mfma_per_wave, double flops_per_mfma, int blocks) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("%-34s %8.1f cyc/MFMA  (%.1f us, %.1f TF/s)\n", name, (double)cy / mfma_per_wave, ms * 1e3,
           mfma_per_wave * 4 * blocks * flops_per_mfma / (ms * 1e-3) / 1e12);
  };
  const int it = 2000;
  run("16x16x4 regs  7 acc, 256 blk", [&]{ hipLaunchKernelGGL((k16<7,false>), dim3(256), dim3(256), 0, 0, d, c, it); }, it*28.0, 2048, 256);
  run("16x16x4 LDS   7 acc, 256 blk", [&]{ hipLaunchKernelGGL((k16<7,true>), dim3(256), dim3(256), 0, 0, d, c, it); }, it*28.0, 2048, 256);
  run("16x16x4 regs  7 acc, 512 blk", [&]{ hipLaunchKernelGGL((k16<7,false>), dim3(512), dim3(256), 0, 0, d, c, it); }, it*28.0, 2048, 512);
  run("16x16x4 regs  2 acc, 256 blk", [&]{ hipLaunchKernelGGL((k16<2,false>), dim3(256), dim3(256), 0, 0, d, c, it); }, it*8.0, 2048, 256);
  run("32x32x2 regs  2 acc, 256 blk", [&]{ hipLaunchKernelGGL((k32<2>), dim3(256), dim3(256), 0, 0, d, c, it); }, it*8.0, 4096, 256);
  run("32x32x2 regs  4 acc, 256 blk", [&]{ hipLaunchKernelGGL((k32<4>), dim3(256), dim3(256), 0, 0, d, c, it); }, it*16.0, 4096, 256);
  return 0;
}
