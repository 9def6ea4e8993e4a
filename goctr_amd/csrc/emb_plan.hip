// emb_plan.hip -- build of the per-batch sparse PLAN of the trainable-embedding extension (emb_train.h, "Round 3").
//
// No reference counterpart: go-ctr trains with frozen embeddings (din.go:161-169, dnn.go:152-154).
// Per batch k of the dataset (model.Train walks the same fixed batches epoch after epoch, model/model.go:96-211) the plan
// lists the batch's (sample, slot) pairs SORTED BY EMBEDDING ROW and the distinct rows ("slots") in ascending owner-major
// order.  Round 3 built it with one global atomicAdd / atomicSub per pair on per-row counters (273 + 277 us per cfg3 batch on
// Zipf-hot rows, and a pair order that depended on the atomics' arrival order).  Round 4: a STABLE SORT of the batch's
// <= B (T + 1) keys by owner-major row index -- the pairs of a row keep their (sample, slot) order, so two builds are byte
// identical -- followed by one head-flag pass, one prefix sum and one fill.  No atomics, no per-batch host read-back (the
// running offsets live on the device), temporaries sized for ONE batch.
//   keys        key[p] = owner-major index of pair p's row (sentinel Vp for pad slots / missing ids), val[p] = b << 12 | t
//   sort        rocprim::radix_sort_pairs over the ceil(log2(Vp + 1)) significant bits (LSD radix: stable)
//   heads       flag[i] = key[i] starts a run of a real row
//   scan + fill scan.h's exclusive prefix sum of the flags; its sink writes pair / pslot / pid, slot_id / slot_off
//   tail        closes the batch: slot_off[n_slots] = n_pairs, advances pair_off / slot_base, tracks the maxima
#include <cstring>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"
#include "emb_plan.h"
#include "scan.h"

namespace goctr {
namespace {

constexpr int PAIR_TBITS = 12;       // = EMB_PAIR_TBITS (emb_train.h): pair code = b << 12 | t

struct KeysArgs {
  const int32_t* ub_ids; const int32_t* item_ids; long long rows;
  int B, T; long long V; int W; long long Vw; long long batch;
};

__global__ __launch_bounds__(256) void emb_plan_keys_kernel(KeysArgs a, unsigned int sentinel, unsigned int* __restrict__ key,
                                                            unsigned int* __restrict__ val) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  const int per = a.T + 1;
  if (p >= (long long)a.B * per) return;
  const int b = (int)(p / per), t = (int)(p - (long long)b * per);
  const long long gr = a.batch * (long long)a.B + b;
  int id = -1;
  if (gr < a.rows) id = t < a.T ? a.ub_ids[gr * a.T + t] : a.item_ids[gr];
  unsigned int k = sentinel;
  if (id >= 0 && id < a.V) {
    if (a.W == 1) k = (unsigned int)id;
    else { const int q = id / a.W; k = (unsigned int)((long long)(id - q * a.W) * a.Vw + q); }     // owner-major: owner = id % W
  }
  key[p] = k;
  val[p] = ((unsigned int)b << PAIR_TBITS) | (unsigned int)t;
}

__global__ __launch_bounds__(256) void emb_plan_heads_kernel(const unsigned int* __restrict__ key, long long n, unsigned int sentinel,
                                                             unsigned int* __restrict__ flag, unsigned long long* __restrict__ n_pairs) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned int k = key[i];
  const bool real = k != sentinel;
  flag[i] = (real && (i == 0 || key[i - 1] != k)) ? 1u : 0u;
  // the sorted list ends with the sentinels: exactly one position is "real and followed by a sentinel or the end"
  if (real && (i + 1 == n || key[i + 1] == sentinel)) *n_pairs = (unsigned long long)(i + 1);
}

// sink of the flags' prefix sum: element i (a sorted pair) with its exclusive count of heads before it
struct FillSink {
  const unsigned int* key; const unsigned int* val; unsigned int sentinel; int W; long long Vw;
  const long long* pair_off; const long long* slot_base; long long batch;      // running offsets (device)
  int* pair; int* pslot; int* pid; int* slot_id; unsigned int* slot_off;
  __device__ __forceinline__ void operator()(long long i, unsigned int head, unsigned int rank) const {
    const unsigned int k = key[i];
    if (k == sentinel) return;
    const long long pb = pair_off[batch], sb = slot_base[batch];
    const unsigned int slot = head ? rank : rank - 1u;
    const int id = W == 1 ? (int)k : (int)(((long long)k % Vw) * W + (long long)k / Vw);
    pair[pb + i] = (int)val[i]; pslot[pb + i] = (int)slot; pid[pb + i] = id;
    if (head) { slot_id[sb + slot] = id; slot_off[sb + batch + slot] = (unsigned int)i; }
  }
};

// totals: [0] pairs of all batches, [1] slots of all batches, [2] max pairs per batch, [3] max slots per batch
__global__ void emb_plan_tail_kernel(const unsigned long long* n_slots, unsigned long long* n_pairs, long long batch, long long* pair_off,
                                     long long* slot_base, unsigned int* slot_off, long long* totals) {
  const long long np = (long long)*n_pairs, ns = (long long)*n_slots;
  slot_off[slot_base[batch] + batch + ns] = (unsigned int)np;
  pair_off[batch + 1] = pair_off[batch] + np;
  slot_base[batch + 1] = slot_base[batch] + ns;
  totals[0] = pair_off[batch + 1]; totals[1] = slot_base[batch + 1];
  if (np > totals[2]) totals[2] = np;
  if (ns > totals[3]) totals[3] = ns;
  *n_pairs = 0;                                  // (the heads kernel of a batch without a single real pair writes nothing)
}

}  // namespace

int emb_plan_build(const EmbPlanSource& src, int B, int T, int W, long long Vw, long long nb, const EmbPlanArrays& out, long long totals_host[4]) {
  Engine& e = engine();
  hipStream_t s = e.stream;
  const long long P = (long long)B * (T + 1);
  const long long Vp = Vw * W;
  GOCTR_CHECK(Vp < 0xFFFFFFFFll && P < (1ll << 31), "embedding plan: vocabulary / batch too large for 32-bit keys");
  const unsigned int sentinel = (unsigned int)Vp;
  unsigned int bits = 1;
  while ((1ull << bits) <= (unsigned long long)Vp) ++bits;
  DevBuf<unsigned int> key_in, key_out, val_in, val_out, flag, tiles;
  DevBuf<unsigned long long> cnt;      // [0] slots of the batch (scan total), [1] pairs of the batch
  DevBuf<long long> totals;
  DevBuf<char> temp;
  if (key_in.alloc((size_t)P, false) || key_out.alloc((size_t)P, false) || val_in.alloc((size_t)P, false) || val_out.alloc((size_t)P, false) ||
      flag.alloc((size_t)P, false) || cnt.alloc(2) || totals.alloc(4)) return -1;
  size_t temp_bytes = 0;
  GOCTR_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, key_in.p, key_out.p, val_in.p, val_out.p, (size_t)P, 0u, bits, s));
  if (temp.alloc(std::max<size_t>(temp_bytes, 16), false)) return -1;
  GOCTR_HIP(hipMemsetAsync(out.pair_off, 0, sizeof(long long), s));
  GOCTR_HIP(hipMemsetAsync(out.slot_base, 0, sizeof(long long), s));
  const dim3 gp((unsigned)cdiv(P, 256));
  for (long long k = 0; k < nb; ++k) {
    const KeysArgs ka{src.ub_ids, src.item_ids, src.rows, B, T, src.V, W, Vw, k};
    hipLaunchKernelGGL(emb_plan_keys_kernel, gp, dim3(256), 0, s, ka, sentinel, key_in.p, val_in.p);
    GOCTR_HIP(hipGetLastError());
    GOCTR_HIP(rocprim::radix_sort_pairs(temp.p, temp_bytes, key_in.p, key_out.p, val_in.p, val_out.p, (size_t)P, 0u, bits, s));
    hipLaunchKernelGGL(emb_plan_heads_kernel, gp, dim3(256), 0, s, key_out.p, P, sentinel, flag.p, cnt.p + 1);
    GOCTR_HIP(hipGetLastError());
    if (exclusive_scan_sink(flag.p, P, tiles, cnt.p, ScanIdentity{},
                            FillSink{key_out.p, val_out.p, sentinel, W, Vw, out.pair_off, out.slot_base, k, out.pair, out.pslot, out.pid,
                                     out.slot_id, out.slot_off})) return -1;
    hipLaunchKernelGGL(emb_plan_tail_kernel, dim3(1), dim3(1), 0, s, cnt.p, cnt.p + 1, k, out.pair_off, out.slot_base, out.slot_off, totals.p);
    GOCTR_HIP(hipGetLastError());
  }
  return totals.download(totals_host, 4);       // (the one synchronisation of the build; the temporaries are released behind it)
}

}  // namespace goctr
