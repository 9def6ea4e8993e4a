// Exclusive prefix sum of 32-bit counters in HBM (three launches: tile sums / scan of the tile sums / apply).
// Used by the dictionary build (corpus.hip) and the sparse embedding update (emb_train.h).
#pragma once
#include "common.h"

namespace goctr {
namespace {

constexpr int SCAN_ITEMS = 16, SCAN_BLOCK = 256, SCAN_TILE = SCAN_ITEMS * SCAN_BLOCK;

__device__ __forceinline__ unsigned int block_exclusive_scan(unsigned int v, unsigned int* total) {
  __shared__ unsigned int wsum[SCAN_BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  unsigned int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
    if (w < wave) base += wsum[w];
    tot += wsum[w];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_tile_sums_kernel(const unsigned int* in, long long n, unsigned int* tile_sum) {
  const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  unsigned int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) s += base + k < n ? in[base + k] : 0u;
  unsigned int tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

// one block walks the tile sums in chunks of SCAN_BLOCK; tile_sum becomes the exclusive scan, total[0] the grand total
__global__ __launch_bounds__(SCAN_BLOCK) void scan_tile_offsets_kernel(unsigned int* tile_sum, long long tiles, unsigned long long* total) {
  unsigned int carry = 0;
  for (long long t0 = 0; t0 < tiles; t0 += SCAN_BLOCK) {
    const long long t = t0 + threadIdx.x;
    const unsigned int v = t < tiles ? tile_sum[t] : 0u;
    unsigned int tot;
    const unsigned int ex = block_exclusive_scan(v, &tot);
    if (t < tiles) tile_sum[t] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_kernel(const unsigned int* in, long long n, const unsigned int* tile_off, unsigned int* out) {
  const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  unsigned int v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = base + k < n ? in[base + k] : 0u; s += v[k]; }
  unsigned int tot;
  unsigned int run = tile_off[blockIdx.x] + block_exclusive_scan(s, &tot);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
}

int exclusive_scan(const unsigned int* in, long long n, unsigned int* out, DevBuf<unsigned int>& tiles_buf,
                   unsigned long long* total_dev) {
  const long long tiles = cdiv(n, SCAN_TILE);
  if (tiles_buf.ensure((size_t)tiles, false)) return -1;
  hipStream_t s = engine().stream;
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, s, in, n, tiles_buf.p);
  hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, tiles_buf.p, tiles, total_dev);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, s, in, n, tiles_buf.p, out);
  GOCTR_HIP(hipGetLastError());
  return 0;
}


}  // namespace
}  // namespace goctr
