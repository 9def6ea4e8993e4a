// huffman.hip -- the Huffman tree of item2vec's hierarchical softmax built WITH the device.
//
// Replaces Dictionary.HuffnamTree + node.GetPath (feature/embedding/corpus/dictionary/huffman.go:23-57, node/node.go:26-43)
// at vocabulary sizes where the host-only builder of w2v.hip (243 ms at V = 10^6, 1.6 s at V = 10^7 on 8 cores: a radix sort, the
// merge loop and the path fill all walking 2 V-entry arrays in random order) is several training passes long.  Division of labour:
//   device  stable radix sort of (count, word) by count                 rocprim::radix_sort_pairs; ties keep word order = the
//                                                                        reference's sort.SliceStable (huffman.go:27-29)
//   host    the two-queue merge, in SORTED-RANK space                    inherently sequential (V - 1 dependent steps), but with
//           leaves numbered by sorted rank every access is a stream: ~5 ns per merge, not a cache miss per merge.  Same rule as
//           w2v.hip's build_huffman: a merged node goes in FRONT of every node of equal value (huffman.go:44-52)
//   device  per leaf (in sorted order: neighbours share their ancestors) the chain length, a prefix sum of the kept path
//           lengths in word order, and the root-first (inner node, code) fill; the paths never exist on the host
// The result is bit-identical to build_huffman's (tests/test_huffman_scale.py).
#include <cstdlib>
#include <cstring>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <chrono>
#include <numeric>

#include "common.h"
#include "huffman.h"
#include "scan.h"

namespace goctr {
namespace {

__global__ __launch_bounds__(256) void huff_iota_kernel(int* idx, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) idx[i] = (int)i;
}

// leaf of sorted rank r: chain length (itself .. root) and the number of path entries GetPath keeps (node.go:39-42)
__global__ __launch_bounds__(256) void huff_len_kernel(const int* __restrict__ parent, const int* __restrict__ order, long long V, int max_depth,
                                                       int* __restrict__ len_r, unsigned int* __restrict__ keep_word) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= V) return;
  int len = 1;
  for (int p = parent[r]; p >= 0; p = parent[p]) ++len;
  len_r[r] = len;
  const int d = len < max_depth ? len : max_depth;
  keep_word[order[r]] = (unsigned int)(d > 0 ? d - 1 : 0);
}

struct OffSink {        // exclusive prefix sums as the 64-bit offsets the item2vec kernels read
  long long* off;
  __device__ __forceinline__ void operator()(long long i, unsigned int, unsigned int rank) const { off[i] = (long long)rank; }
};
__global__ void huff_off_tail_kernel(const unsigned long long* total, long long* off, long long V) { off[V] = (long long)*total; }

// root-first entries j = 0 .. keep-1 of word order[r]: (chain[len-1-j] - V, code[chain[len-2-j]]), written while walking up
__global__ __launch_bounds__(256) void huff_fill_kernel(const int* __restrict__ parent, const unsigned char* __restrict__ code,
                                                        const int* __restrict__ order, const int* __restrict__ len_r, long long V,
                                                        const long long* __restrict__ off, int* __restrict__ nodes, unsigned char* __restrict__ codes) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= V) return;
  const int word = order[r], len = len_r[r];
  const long long o = off[word];
  const int keep = (int)(off[word + 1] - o);
  int prev = (int)r, p = parent[r];
  for (int q = 1; q < len; ++q) {
    const int j = len - 1 - q;
    if (j < keep) { nodes[o + j] = p - (int)V; codes[o + j] = code[prev]; }
    prev = p; p = parent[p];
  }
}

}  // namespace

int huffman_build_device(const long long* counts_host, int64_t V, int max_depth, DevBuf<long long>& off, DevBuf<int>& nodes,
                         DevBuf<unsigned char>& codes, long long* total_out, double parts_ms[4]) {
  Engine& e = engine();
  hipStream_t s = e.stream;
  GOCTR_CHECK(V >= 1 && V <= 0x3fffffff, "huffman_build_device: V = %lld out of range", (long long)V);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const auto t0 = now();
  // ---- device: stable sort by count
  DevBuf<unsigned long long> key_in, key_out;
  DevBuf<int> idx_in, order;
  DevBuf<char> temp;
  if (key_in.alloc((size_t)V, false) || key_out.alloc((size_t)V, false) || idx_in.alloc((size_t)V, false) || order.alloc((size_t)V, false)) return -1;
  GOCTR_HIP(hipMemcpyAsync(key_in.p, counts_host, sizeof(long long) * (size_t)V, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(huff_iota_kernel, dim3((unsigned)cdiv(V, 256)), dim3(256), 0, s, idx_in.p, (long long)V);
  GOCTR_HIP(hipGetLastError());
  long long mx = 0;
  for (int64_t i = 0; i < V; ++i) { GOCTR_CHECK(counts_host[i] >= 0, "huffman_build_device: negative count"); mx = std::max(mx, counts_host[i]); }
  unsigned int bits = 1;
  while (bits < 63 && (mx >> bits) != 0) ++bits;
  size_t temp_bytes = 0;
  GOCTR_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, key_in.p, key_out.p, idx_in.p, order.p, (size_t)V, 0u, bits, s));
  if (temp.alloc(std::max<size_t>(temp_bytes, 16), false)) return -1;
  GOCTR_HIP(rocprim::radix_sort_pairs(temp.p, temp_bytes, key_in.p, key_out.p, idx_in.p, order.p, (size_t)V, 0u, bits, s));
  std::vector<long long> sval((size_t)V);
  GOCTR_HIP(hipMemcpyAsync(sval.data(), key_out.p, sizeof(long long) * (size_t)V, hipMemcpyDeviceToHost, s));
  GOCTR_HIP(hipStreamSynchronize(s));
  const auto t1 = now();
  // ---- host: the two-queue merge in sorted-rank space (leaf r = rank r, merged node k = V + k)
  const int64_t total = 2 * V - 1;
  std::vector<int> parent((size_t)total, -1);
  std::vector<unsigned char> code((size_t)total, 0);
  {
    std::vector<long long> mval((size_t)std::max<int64_t>(V - 1, 1));
    std::vector<int> mq((size_t)V);
    std::vector<long long> run_val; std::vector<int> run_beg, run_end;
    run_val.reserve(1 << 16); run_beg.reserve(1 << 16); run_end.reserve(1 << 16);
    size_t rfront = 0;
    int mq_n = 0;
    int64_t lq = 0;
    for (int64_t k = 0; k + 1 < V; ++k) {
      int pick[2];
      for (int t = 0; t < 2; ++t) {
        const bool have_leaf = lq < V, have_m = rfront < run_val.size();
        const bool take_m = have_leaf && have_m ? run_val[rfront] <= sval[(size_t)lq] : have_m;
        if (take_m) {
          pick[t] = mq[--run_end[rfront]];
          if (rfront + 1 == run_val.size()) mq_n = run_end[rfront];          // (front run == last run: it is a plain stack)
          if (run_end[rfront] == run_beg[rfront]) ++rfront;
        } else {
          pick[t] = (int)lq++;
        }
      }
      const int id = (int)(V + k);
      const long long v = (pick[0] < V ? sval[(size_t)pick[0]] : mval[(size_t)(pick[0] - V)]) +
                          (pick[1] < V ? sval[(size_t)pick[1]] : mval[(size_t)(pick[1] - V)]);
      mval[(size_t)k] = v;
      code[(size_t)pick[0]] = 0; code[(size_t)pick[1]] = 1;
      parent[(size_t)pick[0]] = id; parent[(size_t)pick[1]] = id;
      if (rfront < run_val.size() && run_val.back() == v) { mq[mq_n++] = id; run_end.back() = mq_n; }
      else { run_val.push_back(v); run_beg.push_back(mq_n); mq[mq_n++] = id; run_end.push_back(mq_n); }
    }
  }
  const auto t2 = now();
  // ---- device: chain lengths, offsets, fill
  DevBuf<int> d_parent, len_r;
  DevBuf<unsigned char> d_code;
  DevBuf<unsigned int> keep_word, tiles;
  DevBuf<unsigned long long> tot;
  if (d_parent.alloc((size_t)total, false) || d_code.alloc((size_t)total, false) || len_r.alloc((size_t)V, false) ||
      keep_word.alloc((size_t)V, false) || tot.alloc(1) || off.alloc((size_t)V + 1, false)) return -1;
  GOCTR_HIP(hipMemcpyAsync(d_parent.p, parent.data(), sizeof(int) * (size_t)total, hipMemcpyHostToDevice, s));
  GOCTR_HIP(hipMemcpyAsync(d_code.p, code.data(), (size_t)total, hipMemcpyHostToDevice, s));
  const dim3 gv((unsigned)cdiv(V, 256));
  hipLaunchKernelGGL(huff_len_kernel, gv, dim3(256), 0, s, d_parent.p, order.p, (long long)V, max_depth, len_r.p, keep_word.p);
  GOCTR_HIP(hipGetLastError());
  if (exclusive_scan_sink(keep_word.p, V, tiles, tot.p, ScanIdentity{}, OffSink{off.p})) return -1;
  hipLaunchKernelGGL(huff_off_tail_kernel, dim3(1), dim3(1), 0, s, tot.p, off.p, (long long)V);
  unsigned long long h_tot = 0;
  if (tot.download(&h_tot, 1)) return -1;           // (synchronises: parent / code host vectors may go)
  GOCTR_CHECK(h_tot < (1ull << 31), "goctr_w2v: Huffman paths with 2^31 entries or more (the Hogwild walk indexes them with 32 bits)");
  if (nodes.alloc(std::max<size_t>((size_t)h_tot, 1), false) || codes.alloc(std::max<size_t>((size_t)h_tot, 1), false)) return -1;
  hipLaunchKernelGGL(huff_fill_kernel, gv, dim3(256), 0, s, d_parent.p, d_code.p, order.p, len_r.p, (long long)V, off.p, nodes.p, codes.p);
  GOCTR_HIP(hipGetLastError());
  GOCTR_HIP(hipStreamSynchronize(s));
  const auto t3 = now();
  if (total_out) *total_out = (long long)h_tot;
  if (parts_ms) { parts_ms[0] = ms(t0, t1); parts_ms[1] = ms(t1, t2); parts_ms[2] = ms(t2, t3); parts_ms[3] = ms(t0, t3); }
  return 0;
}

}  // namespace goctr
