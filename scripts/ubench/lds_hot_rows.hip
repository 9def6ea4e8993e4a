// LDS staging of hot embedding rows in the CTR gather (north_star names it; VERDICT r4 missing 3) -- measured in isolation.
// The gather + pooling part of attn_fwd for one batch: B samples x (T + 1) ids -> rows of D floats from a [V, D] table, summed per
// sample.  Ids come from a file the driver (scripts/lds_hot_rows.py) writes with bench.py's own generator (Zipf(1.05) mod V, 20 %
// pad), frequency-ranked so that the HOT most frequent rows are ids < HOT.
//   direct   one wavefront per sample, 16 rows in flight per pass, every row from L2 / HBM (what attn_fwd_kernel does)
//   staged   one persistent 256-thread workgroup per CU: the HOT hottest rows into LDS first (LDS contents do not outlive a launch:
//            the fill is paid per workgroup per launch), then its B / grid samples, a row from LDS when id < HOT
// usage: lds_hot_rows ids.bin B T V D HOT    (D = 16 or 64)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int D, class Ld>
__device__ __forceinline__ void pool_sample(Ld ld, const int* __restrict__ ids, int T1, int V, int lane, float* out) {
  constexpr int LPR = D / 4, RPP = 64 / LPR;                 // lanes per row, rows per pass
  const int dl = lane % LPR, rl = lane / LPR;
  f4 acc = {0, 0, 0, 0};
  for (int t0 = 0; t0 < T1; t0 += 64) {
    const int my = t0 + lane < T1 ? ids[t0 + lane] : -1;
    for (int p = 0; p * RPP < 64 && t0 + p * RPP < T1; p += 4) {
      f4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int slot = (p + u) * RPP + rl;
        const int id = slot < 64 ? __shfl(my, slot, 64) : -1;
        x[u] = (id >= 0 && id < V) ? ld(id, dl) : f4{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += x[u];
    }
  }
  for (int o = LPR; o < 64; o <<= 1) { acc[0] += __shfl_xor(acc[0], o, 64); acc[1] += __shfl_xor(acc[1], o, 64); acc[2] += __shfl_xor(acc[2], o, 64); acc[3] += __shfl_xor(acc[3], o, 64); }
  if (rl == 0) *reinterpret_cast<f4*>(out + 4 * dl) = acc;
}

template <int D>
__global__ __launch_bounds__(256) void gather_direct(const float* tab, const int* ids, int B, int T1, int V, float* out) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  auto ld = [&](int id, int dl) { return *reinterpret_cast<const f4*>(tab + (size_t)id * D + 4 * dl); };
  if (b < B) pool_sample<D>(ld, ids + (size_t)b * T1, T1, V, threadIdx.x & 63, out + (size_t)b * D);
}
template <int D>
__global__ __launch_bounds__(256) void gather_staged(const float* tab, const int* ids, int B, int T1, int V, int HOT, float* out) {
  extern __shared__ __attribute__((aligned(16))) float hot[];
  for (int i = threadIdx.x; i < HOT * D / 4; i += 256) reinterpret_cast<f4*>(hot)[i] = reinterpret_cast<const f4*>(tab)[i];
  __syncthreads();
  auto ld = [&](int id, int dl) {          // (two typed accesses: the LDS branch is a ds_read, not a flat load)
    return id < HOT ? *reinterpret_cast<const f4*>(hot + (size_t)id * D + 4 * dl) : *reinterpret_cast<const f4*>(tab + (size_t)id * D + 4 * dl);
  };
  for (int b = blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += gridDim.x * 4)
    pool_sample<D>(ld, ids + (size_t)b * T1, T1, V, threadIdx.x & 63, out + (size_t)b * D);
}

int main(int argc, char** argv) {
  if (argc < 7) { fprintf(stderr, "usage: %s ids.bin B T V D HOT\n", argv[0]); return 2; }
  const int B = atoi(argv[2]), T1 = atoi(argv[3]) + 1, V = atoi(argv[4]), D = atoi(argv[5]), HOT = atoi(argv[6]);
  std::vector<int> ids((size_t)B * T1);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(ids.data(), 4, ids.size(), f) != ids.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  fclose(f);
  float *tab, *o1, *o2; int* dids;
  hipMalloc(&tab, (size_t)(V + 1) * D * 4); hipMalloc(&o1, (size_t)B * D * 4); hipMalloc(&o2, (size_t)B * D * 4); hipMalloc(&dids, ids.size() * 4);
  std::vector<float> h((size_t)(V + 1) * D);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
  hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const size_t lds = (size_t)HOT * D * 4;
  if (D == 16) hipFuncSetAttribute(reinterpret_cast<const void*>(gather_staged<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  else hipFuncSetAttribute(reinterpret_cast<const void*>(gather_staged<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](int which, int wgs_per_cu) {
    const int iters = 200;
    for (int it = -20; it < iters; ++it) {
      if (it == 0) hipEventRecord(e0, 0);
      if (which == 0) { if (D == 16) gather_direct<16><<<(B + 3) / 4, 256>>>(tab, dids, B, T1, V, o1); else gather_direct<64><<<(B + 3) / 4, 256>>>(tab, dids, B, T1, V, o1); }
      else { if (D == 16) gather_staged<16><<<cus * wgs_per_cu, 256, lds>>>(tab, dids, B, T1, V, HOT, o2); else gather_staged<64><<<cus * wgs_per_cu, 256, lds>>>(tab, dids, B, T1, V, HOT, o2); }
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
  };
  const float td = run(0, 1), ts1 = run(1, 1), ts2 = lds * 2 <= 160 * 1024 ? run(1, 2) : -1.f;
  std::vector<float> a((size_t)B * D), b((size_t)B * D);
  hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0; for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
  long long hits = 0, reads = 0; for (int x : ids) { if (x >= 0 && x < V) { ++reads; hits += x < HOT; } }
  printf("B %d T+1 %d V %d D %d HOT %d (%zu KB of LDS per workgroup): hot-set share of the reads %.3f | direct %.2f us | staged, 1 workgroup per CU %.2f us"
         " | staged, 2 per CU %.2f us | results %s\n", B, T1, V, D, HOT, lds / 1024, (double)hits / reads, td, ts1, ts2, bad ? "DIFFER" : "equal");
  return bad != 0;
}
