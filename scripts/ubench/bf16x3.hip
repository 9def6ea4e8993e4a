// Is a 3-way bf16 split (6 bf16 MFMAs per product, fp32 accumulate) a usable stand-in for v_mfma_f32_16x16x4_f32 on
// gfx950?  Measures (a) the error of both against a float64 reference on the same operands and (b) issue cycles per
// 16-deep k-slice.  Context: DESIGN.md 4.1 — the f32-input MFMA blocks its SIMD, the bf16 one does not.
//   hipcc --offload-arch=gfx950 -O3 -o bf16x3 bf16x3.hip && ./bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));   // gfx950: v_mfma_f32_16x16x32_bf16 takes 8 bf16 per lane

__device__ __forceinline__ unsigned short bf16_rne(float x) {
  unsigned int u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
  hi = bf16_rne(x);
  const float r1 = x - bf16_f32(hi);
  mid = bf16_rne(r1);
  const float r2 = r1 - bf16_f32(mid);
  lo = bf16_rne(r2);
}

// C[16,16] = A[16,K] . B[K,16], one wavefront; out32 = f32 MFMA, out6 = bf16 split with 6 products, out3 = 3 products
__global__ __launch_bounds__(64) void gemm_kernel(const float* A, const float* B, int K, float* out32, float* out6, float* out3) {
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  f4 c32{0, 0, 0, 0}, chh{0, 0, 0, 0}, cm{0, 0, 0, 0}, cs{0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    // f32: four 16x16x4 steps; lane holds A[i][k0 + 4 s + q], B[k0 + 4 s + q][i]
    for (int s = 0; s < 4; ++s)
      c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k0 + 4 * s + q], B[(k0 + 4 * s + q) * 16 + i], c32, 0, 0, 0);
    // bf16: one 16x16x16 step per product; lane holds A[i][k0 + 4 q + e], B[k0 + 4 q + e][i], e = 0..3
    s4 ah, am, al, bh, bm, bl;
    for (int e = 0; e < 4; ++e) {
      unsigned short h, m, l;
      split3(A[i * K + k0 + 4 * q + e], h, m, l); ah[e] = (short)h; am[e] = (short)m; al[e] = (short)l;
      split3(B[(k0 + 4 * q + e) * 16 + i], h, m, l); bh[e] = (short)h; bm[e] = (short)m; bl[e] = (short)l;
    }
    cs = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, cs, 0, 0, 0);
    cs = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, cs, 0, 0, 0);
    cs = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, bm, cs, 0, 0, 0);
    cm = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, bh, cm, 0, 0, 0);
    cm = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bm, cm, 0, 0, 0);
    chh = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, chh, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) {                       // C[4 q + r][i]
    out32[(4 * q + r) * 16 + i] = c32[r];
    out6[(4 * q + r) * 16 + i] = (cs[r] + cm[r]) + chh[r];
    out3[(4 * q + r) * 16 + i] = cm[r] + chh[r];
  }
}

// issue rate: per 16-deep k-slice 4 f32 MFMAs vs 6 bf16 MFMAs (operands in registers), 4 wavefronts per workgroup
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, unsigned long long* cyc, int iters) {
  f4 acc[6];
  for (int t = 0; t < 6; ++t) acc[t] = f4{0, 0, 0, 0};
  const float a = threadIdx.x * 0.001f;
  s4 h{1, 2, 3, 4};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc[t], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(h, h, acc[t], 0, 0, 0);
    } else {
      b8 w;
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = (__bf16)(a + e);
#pragma unroll
      for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, w, acc[t], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < 6; ++t) s += acc[t][0] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  const int K = 208;
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<float> A(16 * K), B(K * 16);
    srand(1 + mode);
    auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
    for (auto& v : A) v = mode == 0 ? rnd() : mode == 1 ? 1.0f / (1.0f + expf(-8.0f * rnd())) : rnd() * expf(12.0f * rnd());
    for (auto& v : B) v = mode == 2 ? rnd() * expf(12.0f * rnd()) : 2.0f * rnd();
    float *dA, *dB, *d32, *d6, *d3;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d32, 1024); hipMalloc(&d6, 1024); hipMalloc(&d3, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gemm_kernel, dim3(1), dim3(64), 0, 0, dA, dB, K, d32, d6, d3);
    float c32[256], c6[256], c3[256];
    hipMemcpy(c32, d32, 1024, hipMemcpyDeviceToHost); hipMemcpy(c6, d6, 1024, hipMemcpyDeviceToHost); hipMemcpy(c3, d3, 1024, hipMemcpyDeviceToHost);
    double e32 = 0, e6 = 0, e3 = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double ref = 0, mag = 0;
      for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 16 + j]; mag += fabs((double)A[i * K + k] * B[k * 16 + j]); }
      e32 = fmax(e32, fabs(c32[i * 16 + j] - ref) / mag); e6 = fmax(e6, fabs(c6[i * 16 + j] - ref) / mag); e3 = fmax(e3, fabs(c3[i * 16 + j] - ref) / mag);
    }
    const char* names[3] = {"uniform(-.5,.5) x uniform(-1,1)", "sigmoid activations x uniform(-1,1)", "12-e-fold dynamic range both sides"};
    printf("K=%d %-38s max |err| / sum|a b|:  f32 MFMA %.2e   bf16 x 6 products %.2e   bf16 x 3 products %.2e\n", K, names[mode], e32, e6, e3);
  }
  float* d; unsigned long long* c; hipMalloc(&d, 1 << 22); hipMalloc(&c, 8);
  const int it = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL((rate_kernel<0>), dim3(256), dim3(256), 0, 0, d, c, it);
      else if (mode == 1) hipLaunchKernelGGL((rate_kernel<1>), dim3(256), dim3(256), 0, 0, d, c, it);
      else hipLaunchKernelGGL((rate_kernel<2>), dim3(256), dim3(256), 0, 0, d, c, it);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    const char* nm[3] = {"f32  16x16x4  x4 (k-depth 16)", "bf16 16x16x16 x6 (k-depth 16)", "bf16 16x16x32 x6 (k-depth 32)"};
    const double kdepth = mode == 2 ? 32 : 16;
    printf("%s: %.1f memtime ticks per loop body, %.1f us for %d bodies on 1024 wavefronts => %.1f ns per 16-deep k-slice per wavefront\n",
           nm[mode], (double)cy / it, ms * 1e3, it, ms * 1e6 / it * 16.0 / kdepth);
  }
  return 0;
}
