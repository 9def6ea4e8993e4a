// Does a kernel that runs straight-line code ONCE (every instruction fetch is a cold I-cache miss served by L2) issue more
// slowly than the same dynamic instruction stream in a loop?  Decides whether the fully unrolled phases of ctr_chain_x3 /
// mlp_chain_kernel should be loops.   hipcc --offload-arch=gfx950 -O3 ifetch.hip -o ifetch && ./ifetch
#include <hip/hip_runtime.h>
#include <cstdio>
#define F1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define F4 F1 F1 F1 F1
#define F16 F4 F4 F4 F4
#define F64 F16 F16 F16 F16
#define F256 F64 F64 F64 F64
#define F1024 F256 F256 F256 F256
#define F4096 F1024 F1024 F1024 F1024
__global__ __launch_bounds__(512) void k_loop(float* out, float a, float b, int iters) {
  float x = threadIdx.x;
  for (int i = 0; i < iters; ++i) { F64 }
  out[blockIdx.x * 512 + threadIdx.x] = x;
}
__global__ __launch_bounds__(512) void k_flat(float* out, float a, float b) {
  float x = threadIdx.x;
  F4096 F1024 F1024 F256                  // 6400 instructions, 51 KB of code
  out[blockIdx.x * 512 + threadIdx.x] = x;
}
// two independent chains per wave (ILP 2)
#define G1 asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(a), "v"(b));
#define G4 G1 G1 G1 G1
#define G16 G4 G4 G4 G4
#define G64 G16 G16 G16 G16
#define G256 G64 G64 G64 G64
#define G1024 G256 G256 G256 G256
__global__ __launch_bounds__(512) void k_loop2(float* out, float a, float b, int iters) {
  float x = threadIdx.x, y = x + 1;
  for (int i = 0; i < iters; ++i) { G16 G16 }
  out[blockIdx.x * 512 + threadIdx.x] = x + y;
}
__global__ __launch_bounds__(512) void k_flat2(float* out, float a, float b) {
  float x = threadIdx.x, y = x + 1;
  G1024 G1024 G1024 G64 G64                // 3200 pairs = 6400 instructions
  out[blockIdx.x * 512 + threadIdx.x] = x + y;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, int threads) {
    for (int i = 0; i < 5; ++i) launch(threads);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) launch(threads);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %4d threads/WG: %7.2f us per launch\n", name, threads, ms * 1000 / 200);
  };
  for (int threads : {256, 512}) {
    run("loop 100 x 64 fma", [&](int t) { hipLaunchKernelGGL(k_loop, dim3(256), dim3(t), 0, 0, out, 1.0001f, 0.5f, 100); }, threads);
    run("flat 6400 fma", [&](int t) { hipLaunchKernelGGL(k_flat, dim3(256), dim3(t), 0, 0, out, 1.0001f, 0.5f); }, threads);
    run("loop 100 x 32 pairs", [&](int t) { hipLaunchKernelGGL(k_loop2, dim3(256), dim3(t), 0, 0, out, 1.0001f, 0.5f, 100); }, threads);
    run("flat 3200 pairs", [&](int t) { hipLaunchKernelGGL(k_flat2, dim3(256), dim3(t), 0, 0, out, 1.0001f, 0.5f); }, threads);
    run("empty loop (0 iters)", [&](int t) { hipLaunchKernelGGL(k_loop, dim3(256), dim3(t), 0, 0, out, 1.0001f, 0.5f, 0); }, threads);
  }
  return 0;
}
