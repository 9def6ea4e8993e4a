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

// A tile is SCAN_SUB sub-tiles of SCAN_BLOCK * 4 items; thread t of sub-tile j owns items 4 t .. 4 t + 3 of it, fetched with
// one 16-byte load (a wavefront reads 1 KiB contiguous).
constexpr int SCAN_SUB = SCAN_ITEMS / 4;

__device__ __forceinline__ uint4 scan_load4(const unsigned int* in, long long i, long long n) {
  if (i + 3 < n) return *reinterpret_cast<const uint4*>(in + i);
  uint4 v{0u, 0u, 0u, 0u};
  if (i < n) v.x = in[i];
  if (i + 1 < n) v.y = in[i + 1];
  if (i + 2 < n) v.z = in[i + 2];
  return v;
}

struct ScanIdentity {
  __device__ __forceinline__ unsigned int operator()(unsigned int v) const { return v; }
};

// map(v) is what gets summed (identity for 0 / 1 flags)
template <class Map>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_tile_sums_kernel(const unsigned int* in, long long n, unsigned int* tile_sum, Map map) {
  const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * 4;
  unsigned int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_SUB; ++j) {
    const uint4 v = scan_load4(in, base + (long long)j * SCAN_BLOCK * 4, n);
    s += map(v.x) + map(v.y) + map(v.z) + map(v.w);
  }
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

// sink(i, value, rank) sees every element once with its exclusive prefix sum
template <class Map, class Sink>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_kernel(const unsigned int* in, long long n, const unsigned int* tile_off, Map map, Sink sink) {
  const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * 4;
  uint4 v[SCAN_SUB];
#pragma unroll
  for (int j = 0; j < SCAN_SUB; ++j) v[j] = scan_load4(in, base + (long long)j * SCAN_BLOCK * 4, n);
  unsigned int carry = tile_off[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_SUB; ++j) {
    const long long i = base + (long long)j * SCAN_BLOCK * 4;
    unsigned int tot;
    unsigned int run = carry + block_exclusive_scan(map(v[j].x) + map(v[j].y) + map(v[j].z) + map(v[j].w), &tot);
    carry += tot;
    if (i < n) sink(i, v[j].x, run);
    run += map(v[j].x);
    if (i + 1 < n) sink(i + 1, v[j].y, run);
    run += map(v[j].y);
    if (i + 2 < n) sink(i + 2, v[j].z, run);
    run += map(v[j].z);
    if (i + 3 < n) sink(i + 3, v[j].w, run);
  }
}

struct ScanStore {
  unsigned int* out;
  __device__ __forceinline__ void operator()(long long i, unsigned int, unsigned int rank) const { out[i] = rank; }
};

template <class Map, class Sink>
int exclusive_scan_sink(const unsigned int* in, long long n, DevBuf<unsigned int>& tiles_buf, unsigned long long* total_dev, Map map, Sink sink) {
  const long long tiles = cdiv(n, SCAN_TILE);
  if (tiles_buf.ensure((size_t)tiles, false)) return -1;
  hipStream_t s = engine().stream;
  hipLaunchKernelGGL((scan_tile_sums_kernel<Map>), dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, s, in, n, tiles_buf.p, map);
  hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, tiles_buf.p, tiles, total_dev);
  hipLaunchKernelGGL((scan_apply_kernel<Map, Sink>), dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, s, in, n, tiles_buf.p, map, sink);
  GOCTR_HIP(hipGetLastError());
  return 0;
}

int exclusive_scan(const unsigned int* in, long long n, unsigned int* out, DevBuf<unsigned int>& tiles_buf,
                   unsigned long long* total_dev) {
  return exclusive_scan_sink(in, n, tiles_buf, total_dev, ScanIdentity{}, ScanStore{out});
}

}  // namespace
}  // namespace goctr
