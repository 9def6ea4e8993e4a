// Does a tile written by workgroup b of launch N arrive sooner in launch N + 1 when its reader runs on the SAME XCD?
// (the cfg3 step hands h0, A0 / dz0 and the weight-gradient slabs from launch to launch: every consumer's first phase is the
// wait for what the producer launch wrote "through other XCDs' L2s"; block b runs on XCD b % 8 -- observed, not promised)
//
//   producer: 256 workgroups x 512 threads, workgroup b stores tile b (KB bytes, plain 16-byte stores), value = f(iteration)
//   consumer: workgroup b reads tile map(b): shift 0 = the tile its own XCD wrote, shift s = tile (b + s) % 256
//             (s = 8: another CU of the same XCD, s = 1: the neighbouring XCD), s_memtime from its first instruction to
//             the data in registers; every word checked against the iteration's value (stale lines would show)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void produce(float* buf, int tile_f, float val) {
  f4* t = reinterpret_cast<f4*>(buf + (size_t)blockIdx.x * tile_f);
  for (int i = threadIdx.x; i < tile_f / 4; i += 512) t[i] = f4{val, val + 1.f, val + 2.f, (float)blockIdx.x};
}

template <int NLD>
__global__ __launch_bounds__(512) void consume(const float* buf, int tile_f, int shift, float val, unsigned long long* cyc, unsigned* bad, float* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int src = ((int)blockIdx.x + shift) % (int)gridDim.x;
  const f4* t = reinterpret_cast<const f4*>(buf + (size_t)src * tile_f);
  f4 v[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int i = threadIdx.x + 512 * k;
    v[k] = i < tile_f / 4 ? t[i] : f4{val, val + 1.f, val + 2.f, (float)src};
  }
  unsigned nb = 0;
#pragma unroll
  for (int k = 0; k < NLD; ++k) nb += (v[k][0] != val) + (v[k][1] != val + 1.f) + (v[k][2] != val + 2.f) + (v[k][3] != (float)src);
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (nb) atomicAdd(bad, nb);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (v[0][0] == -12345.f) sink[threadIdx.x] = v[0][1];
}

int main(int argc, char** argv) {
  const int blocks = 256, iters = argc > 1 ? atoi(argv[1]) : 200;
  float* d; unsigned long long* call; unsigned* bad; float* sink;
  hipMalloc(&d, (size_t)blocks * 65536 * 4); hipMalloc(&call, (size_t)iters * blocks * 8); hipMalloc(&bad, 4); hipMalloc(&sink, 4096);
  hipMemset(bad, 0, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  printf("tile KB | shift | consumer wait (s_memtime ticks: median over blocks and iterations, p90) | pair us (events) | stale words\n");
  for (int kb : {16, 32, 64}) {
    const int tile_f = kb * 256;
    for (int shift : {0, 8, 1, 3, 0, 1}) {
      std::vector<unsigned long long> all((size_t)iters * blocks);
      hipStream_t st; hipStreamCreate(&st);
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);          // (replayed back to back like the product's step graphs)
      for (int it = 0; it < iters; ++it) {
        const float val = (float)(it * 7 + kb);
        unsigned long long* c = call + (size_t)it * blocks;
        hipLaunchKernelGGL(produce, dim3(blocks), dim3(512), 0, st, d, tile_f, val);
        if (kb == 16) hipLaunchKernelGGL((consume<2>), dim3(blocks), dim3(512), 0, st, d, tile_f, shift, val, c, bad, sink);
        else if (kb == 32) hipLaunchKernelGGL((consume<4>), dim3(blocks), dim3(512), 0, st, d, tile_f, shift, val, c, bad, sink);
        else hipLaunchKernelGGL((consume<8>), dim3(blocks), dim3(512), 0, st, d, tile_f, shift, val, c, bad, sink);
      }
      hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphLaunch(ge, st); hipStreamSynchronize(st);
      hipEventRecord(a, st); hipGraphLaunch(ge, st);
      hipEventRecord(b, st); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      hipMemcpy(all.data(), call, all.size() * 8, hipMemcpyDeviceToHost);
      all.erase(all.begin(), all.begin() + (size_t)(iters / 4) * blocks);
      std::sort(all.begin(), all.end());
      unsigned nbad; hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost);
      printf("%7d | %5d | %6llu %6llu | %7.2f | %u\n", kb, shift, all[all.size() / 2], all[all.size() * 9 / 10],
             ms * 1e3 / iters, nbad);
      hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
    }
  }
  return 0;
}
