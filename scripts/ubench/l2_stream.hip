// How fast can ONE CU pull an L2-resident weight image when all 256 CUs do it at once?
// (the chain kernel streams ~280 KB of weights per workgroup; this bounds its DMA phases)
//   variants: LDS-DMA (global_load_lds_dwordx4) vs register loads (global_load_dwordx4), 4/8/16 waves per WG,
//             same region for every WG (L2 hits) vs private region per WG
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int WAVES, bool DMA>
__global__ __launch_bounds__(WAVES * 64) void stream(const float* __restrict__ src, size_t wg_stride_f, int region_f, int iters,
                                                     float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* base = src + (size_t)blockIdx.x * wg_stride_f;
  const int nchunks = region_f / 256;   // 1 KiB chunks
  f4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (DMA) {
      for (int c = wave; c < nchunks; c += WAVES)
        __builtin_amdgcn_global_load_lds(base + c * 256 + lane * 4, (__attribute__((address_space(3))) void*)(sm + (c & 63) * 256), 16, 0, 0);
      __syncthreads();
    } else {
#pragma unroll 8
      for (int c = wave; c < nchunks; c += WAVES) acc += *reinterpret_cast<const f4*>(base + c * 256 + lane * 4);
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (DMA) acc = *reinterpret_cast<f4*>(sm + lane * 4);
  out[blockIdx.x * WAVES * 64 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void touch(float* p, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] += 1.0f; }

// first-touch variant: one pass over `region_f` floats, chunks visited from a per-WG rotated start
template <bool ROT>
__global__ __launch_bounds__(256) void first_touch(const float* __restrict__ src, int region_f, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nchunks = region_f / 256;
  const int steps = (nchunks + 3) / 4;
  const int rot = ROT ? blockIdx.x % steps : 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int j = 0; j < steps; ++j) {
    int jj = j + rot; jj = jj >= steps ? jj - steps : jj;
    const int c = wave + 4 * jj;
    if (c < nchunks)
      __builtin_amdgcn_global_load_lds(src + c * 256 + lane * 4, (__attribute__((address_space(3))) void*)(sm + (c & 63) * 256), 16, 0, 0);
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + tid] = sm[tid];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int region_f = 65536;           // 256 KiB per pass
  const int blocks = 256, iters = 20;
  float* d; float* o; unsigned long long* c;
  hipMalloc(&d, (size_t)blocks * region_f * 4); hipMemset(d, 0, (size_t)blocks * region_f * 4);
  hipMalloc(&o, 1 << 22); hipMalloc(&c, 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    const double bytes = (double)region_f * 4 * iters;
    printf("%-44s %6.1f B/cyc/CU  (%8.1f us, aggregate %.2f TB/s)\n", name, bytes / (double)cy, ms * 1e3, bytes * blocks / (ms * 1e-3) / 1e12);
  };
#define RUN(W, D, SHARED) run(#W " waves " #D " " #SHARED, [&] { \
    hipFuncSetAttribute((const void*)stream<W, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
    hipLaunchKernelGGL((stream<W, D>), dim3(blocks), dim3(W * 64), 65536, 0, d, SHARED ? 0 : (size_t)region_f, region_f, iters, o, c); })
  RUN(4, true, 1); RUN(8, true, 1); RUN(16, true, 1);
  RUN(4, false, 1); RUN(8, false, 1); RUN(16, false, 1);
  RUN(4, true, 0); RUN(16, true, 0); RUN(4, false, 0); RUN(16, false, 0);
  unsigned long long* cb; hipMalloc(&cb, 8 * 256);
  hipFuncSetAttribute((const void*)first_touch<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)first_touch<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rot = 0; rot < 2; ++rot)
    for (int kb : {16, 64, 256}) {
      const int rf = kb * 256;
      unsigned long long h[256]; double tot = 0; unsigned long long mx = 0, mn = ~0ull;
      for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(touch, dim3((rf + 255) / 256), dim3(256), 0, 0, d, rf);
        if (rot) hipLaunchKernelGGL(first_touch<true>, dim3(256), dim3(256), 65536, 0, d, rf, o, cb);
        else hipLaunchKernelGGL(first_touch<false>, dim3(256), dim3(256), 65536, 0, d, rf, o, cb);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, cb, sizeof h, hipMemcpyDeviceToHost);
      for (int i = 0; i < 256; ++i) { tot += h[i]; mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; }
      printf("first touch %3d KiB rot=%d: avg %7.0f cyc (min %llu max %llu) => %.1f B/cyc/CU\n", kb, rot, tot / 256, mn, mx, kb * 1024.0 / (tot / 256));
    }
  return 0;
}
