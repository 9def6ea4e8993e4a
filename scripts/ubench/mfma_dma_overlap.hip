// Does an L2 -> LDS weight stream overlap with a register-fed MFMA phase on the same CU?
// 8 wavefronts per workgroup: 0..3 issue NM independent v_mfma_f32_16x16x4_f32, 4..7 copy KB KiB of an
// L2-resident image into LDS (LDS-DMA); one barrier at the end.  Reported: cycles of wave 0 per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool MF, bool DMA, int MODE>
__global__ __launch_bounds__(512, 1) void k(const float* __restrict__ src, int nfl, int nm, float* out, unsigned long long* cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f4 acc[7];
  for (int t = 0; t < 7; ++t) acc[t] = f4{0, 0, 0, 0};
  float b = tid * 0.001f;
  typedef float f16v __attribute__((ext_vector_type(16)));
  typedef short s4 __attribute__((ext_vector_type(4)));
  f16v big[4];
  for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) big[t][j] = 0;
  s4 hb = {(short)tid, 1, 2, 3};
  f4 ld = {0, 0, 0, 0};
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 dacc[7];
  for (int t = 0; t < 7; ++t) dacc[t] = d4{0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
    if (wave < 4) {
      if (MODE == 2) __builtin_amdgcn_s_sleep(8);   // ~512 cycles head start for the loaders
      if (MF && MODE == 4) {          // 32x32x2 f32: 64 cycles each, half as many
        for (int it = 0; it < nm / 8; ++it)
#pragma unroll
          for (int t = 0; t < 4; ++t) big[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, big[t], 0, 0, 0);
      } else if (MF && (MODE == 8 || MODE == 9)) {   // f64 16x16x4 (64 cycles each): half as many
        for (int it = 0; it < nm / 56; ++it)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 7; ++t) dacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b, (double)b, dacc[t], 0, 0, 0);
      } else if (MF && MODE == 5) {   // bf16 16x16x16: 8 passes like the f32 16x16x4
        for (int it = 0; it < nm / 28; ++it)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 7; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hb, hb, acc[t], 0, 0, 0);
      } else if (MF)
        for (int it = 0; it < nm / 28; ++it)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 7; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc[t], 0, 0, 0);
    } else if (DMA) {
      if (MODE == 1) __builtin_amdgcn_s_setprio(3);
      const int nchunks = nfl / 256;
      if (MODE == 9) {   // pure VALU beside the f64 MFMAs
        float x0 = b, x1 = b + 1, x2 = b + 2, x3 = b + 3;
        for (int it = 0; it < nm * 2; ++it) {
          x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f);
          x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f);
        }
        ld[0] += x0 + x1 + x2 + x3;
      } else if (MODE == 6) {   // pure VALU work in the partner wavefronts: nm*8 dependent-free FMAs
        float x0 = b, x1 = b + 1, x2 = b + 2, x3 = b + 3;
        for (int it = 0; it < nm * 2; ++it) {
          x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f);
          x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f);
        }
        ld[0] += x0 + x1 + x2 + x3;
      } else if (MODE == 7) {   // LDS reads in the partner wavefronts
        for (int it = 0; it < nm; ++it) ld += *reinterpret_cast<f4*>(sm + ((it * 64 + lane) & 8191) * 4);
      } else if (MODE == 3) {
#pragma unroll 17
        for (int c = wave - 4; c < nchunks; c += 4) ld += *reinterpret_cast<const f4*>(src + c * 256 + lane * 4);
      } else
      for (int c = wave - 4; c < nchunks; c += 4)
        __builtin_amdgcn_global_load_lds(src + c * 256 + lane * 4, (__attribute__((address_space(3))) void*)(sm + c * 256), 16, 0, 0);
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = sm[tid];
  for (int t = 0; t < 7; ++t) s += acc[t][0] + acc[t][3];
  for (int t = 0; t < 4; ++t) s += big[t][0] + big[t][7];
  s += ld[0] + ld[1] + ld[2] + ld[3];
  for (int t = 0; t < 7; ++t) s += (float)(dacc[t][0] + dacc[t][3]);
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float* d; float* o; unsigned long long* c;
  hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22); hipMalloc(&o, 1 << 22); hipMalloc(&c, 8);
  hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
  const int reps = 8;
  auto run = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(ea); launch(); hipEventRecord(eb); hipEventSynchronize(eb);
    float ms; hipEventElapsedTime(&ms, ea, eb);
    unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("%-40s %8.0f cyc/phase  (%.1f us total)\n", name, (double)cy / reps, ms * 1e3);
  };
#define RUNM(MF, DMA, KB, NM, BLK, MODE) do { char nm_[96]; snprintf(nm_, 96, "mfma=%d dma=%d %3d KiB %3d MFMA %3d WG mode %d", MF, DMA, KB, NM, BLK, MODE); \
    hipFuncSetAttribute((const void*)k<MF, DMA, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
    run(nm_, [&] { hipLaunchKernelGGL((k<MF, DMA, MODE>), dim3(BLK), dim3(512), 131072, 0, d, KB * 256, NM, o, c, reps); }); } while (0)
#define RUN(MF, DMA, KB, NM, BLK) RUNM(MF, DMA, KB, NM, BLK, 0)
  RUN(true, false, 65, 140, 256);
  RUN(false, true, 65, 140, 256);
  RUN(true, true, 65, 140, 256);
  RUN(true, true, 65, 280, 256);
  RUN(true, true, 32, 140, 256);
  RUN(true, true, 65, 140, 32);
  RUN(false, true, 65, 140, 32);
  RUN(true, true, 65, 140, 1);
  RUN(false, true, 65, 140, 1);
  RUNM(true, true, 65, 140, 256, 1);
  RUNM(true, true, 65, 140, 256, 2);
  RUNM(true, false, 65, 140, 256, 2);
  RUNM(true, true, 65, 140, 256, 3);
  RUNM(false, true, 65, 140, 256, 3);
  RUNM(true, false, 65, 140, 256, 4);
  RUNM(true, true, 65, 140, 256, 4);
  RUNM(true, false, 65, 140, 256, 5);
  RUNM(true, true, 65, 140, 256, 5);
  RUNM(false, true, 65, 140, 256, 6);
  RUNM(true, true, 65, 140, 256, 6);
  RUNM(false, true, 65, 140, 256, 7);
  RUNM(true, true, 65, 140, 256, 7);
  RUNM(true, false, 65, 140, 256, 8);
  RUNM(true, true, 65, 140, 256, 8);
  RUNM(true, true, 65, 140, 256, 9);
  return 0;
}
