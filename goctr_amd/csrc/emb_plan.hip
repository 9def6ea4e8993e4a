// emb_plan.hip -- build of the per-batch sparse PLAN of the trainable-embedding extension (emb_train.h, "Round 3").
//
// No reference counterpart: go-ctr trains with frozen embeddings (din.go:161-169, dnn.go:152-154).
// Per batch k of the dataset (model.Train walks the same fixed batches epoch after epoch, model/model.go:96-211) the plan
// lists the batch's (sample, slot) pairs SORTED BY EMBEDDING ROW and the distinct rows ("slots") in ascending owner-major
// order.  Round 3 built it with one global atomicAdd / atomicSub per pair on per-row counters (273 + 277 us per cfg3 batch on
// Zipf-hot rows, and a pair order that depended on the atomics' arrival order).  Round 4: ONE STABLE SORT over the keys of
// many batches at once, key = (batch, owner-major row index) -- the pairs of a row keep their (sample, slot) order, so two
// builds are byte identical -- followed by one head-flag pass, one prefix sum, one offsets pass and one fill.  No atomics,
// no per-batch launches or read-backs (sorting a 420 k-key batch on its own is launch-latency: ~20 launches of a few us; 32
// batches in one sort are bandwidth), temporaries bounded by 1 GiB (EMB_PLAN_TMP_MB).
//   keys     key[p] = batch-in-chunk << bits | owner-major index of pair p's row (sentinel Vp for pad slots / missing ids:
//            sorts behind the batch's real rows), val[p] = b << 12 | t
//   sort     rocprim::radix_sort_pairs over the significant bits (stable); batch kb then occupies keys [kb P, (kb + 1) P)
//   heads    flag[i] = key[i] starts a run of a real row; npairs[kb]
//   scan     scan.h's exclusive prefix sum of the flags -> rank[i]
//   offsets  one thread appends the chunk's batches to pair_off / slot_base, closes their run-start lists, tracks the maxima
//   fill     pair / pslot / pid per sorted pair, slot_id / slot_off per head
#include <cstdlib>
#include <cstring>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"
#include "emb_plan.h"
#include "scan.h"

namespace goctr {
namespace {

constexpr int PAIR_TBITS = 12;       // = EMB_PAIR_TBITS (emb_train.h): pair code = b << 12 | t

constexpr long long EMB_PLAN_TMP_MB = 1024;     // budget of the sort's temporaries

struct KeysArgs {
  const int32_t* ub_ids; const int32_t* item_ids; long long rows;
  int B, T; long long V; int W; long long Vw; long long batch0;     // batch0: first batch of the chunk
  long long P; unsigned int bits, sentinel;                          // P = B (T + 1) keys per batch; key = batch-in-chunk << bits | row index
};

// key of pair p of the chunk: (batch within the chunk, owner-major index of the pair's row); pad slots / missing ids get the
// sentinel row Vp, which sorts behind every real row of its batch
__global__ __launch_bounds__(256) void emb_plan_keys_kernel(KeysArgs a, long long n, unsigned int* __restrict__ key, unsigned int* __restrict__ val) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int per = a.T + 1;
  const long long kb = p / a.P, q = p - kb * a.P;
  const int b = (int)(q / per), t = (int)(q - (long long)b * per);
  const long long gr = (a.batch0 + kb) * (long long)a.B + b;
  int id = -1;
  if (gr < a.rows) id = t < a.T ? a.ub_ids[gr * a.T + t] : a.item_ids[gr];
  unsigned int k = a.sentinel;
  if (id >= 0 && id < a.V) {
    if (a.W == 1) k = (unsigned int)id;
    else { const int o = id / a.W; k = (unsigned int)((long long)(id - o * a.W) * a.Vw + o); }     // owner-major: owner = id % W
  }
  key[p] = ((unsigned int)kb << a.bits) | k;
  val[p] = ((unsigned int)b << PAIR_TBITS) | (unsigned int)t;
}

// over the sorted keys: flag[i] = "starts the run of a real row" (the batch is part of the key: a new batch starts a new run),
// npairs[kb] = real pairs of batch kb (its sentinels sit at the end of its P keys)
__global__ __launch_bounds__(256) void emb_plan_heads_kernel(const unsigned int* __restrict__ key, long long n, long long P, unsigned int bits,
                                                             unsigned int sentinel, unsigned int* __restrict__ flag, unsigned int* __restrict__ npairs) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned int k = key[i], mask = (1u << bits) - 1u;
  const bool real = (k & mask) != sentinel;
  flag[i] = (real && (i == 0 || key[i - 1] != k)) ? 1u : 0u;
  const long long kb = i / P, j = i - kb * P;
  if (real && (j + 1 == P || (key[i + 1] & mask) == sentinel)) npairs[kb] = (unsigned int)(j + 1);
}

// one thread: the chunk's batches appended to the running offsets; totals = {pairs, slots, max pairs, max slots}
__global__ void emb_plan_offsets_kernel(const unsigned int* rank, const unsigned int* flag, const unsigned int* npairs, long long P, long long kcount,
                                        long long batch0, long long* pair_off, long long* slot_base, unsigned int* slot_off, long long* totals) {
  for (long long kb = 0; kb < kcount; ++kb) {
    const long long k = batch0 + kb;
    const long long r0 = rank[kb * P];
    const long long r1 = kb + 1 < kcount ? (long long)rank[(kb + 1) * P] : (long long)rank[kcount * P - 1] + (long long)flag[kcount * P - 1];
    const long long np = npairs[kb], ns = r1 - r0;
    pair_off[k + 1] = pair_off[k] + np;
    slot_base[k + 1] = slot_base[k] + ns;
    slot_off[slot_base[k] + k + ns] = (unsigned int)np;                // the closing entry of the batch's run starts
    if (np > totals[2]) totals[2] = np;
    if (ns > totals[3]) totals[3] = ns;
  }
  totals[0] = pair_off[batch0 + kcount]; totals[1] = slot_base[batch0 + kcount];
}

struct FillArgs {
  const unsigned int* key; const unsigned int* val; const unsigned int* flag; const unsigned int* rank;
  long long n, P, batch0; unsigned int bits, sentinel; int W; long long Vw;
  const long long* pair_off; const long long* slot_base;
  int* pair; int* pslot; int* pid; int* slot_id; unsigned int* slot_off;
};
__global__ __launch_bounds__(256) void emb_plan_fill_kernel(FillArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const unsigned int k = a.key[i] & ((1u << a.bits) - 1u);
  if (k == a.sentinel) return;
  const long long kb = i / a.P, j = i - kb * a.P, kk = a.batch0 + kb;
  const unsigned int head = a.flag[i];
  const unsigned int slot = a.rank[i] - a.rank[kb * a.P] - (head ? 0u : 1u);       // slot inside the batch
  const long long pb = a.pair_off[kk], sb = a.slot_base[kk];
  const int id = a.W == 1 ? (int)k : (int)(((long long)k % a.Vw) * a.W + (long long)k / a.Vw);
  a.pair[pb + j] = (int)a.val[i]; a.pslot[pb + j] = (int)slot; a.pid[pb + j] = id;
  if (head) { a.slot_id[sb + slot] = id; a.slot_off[sb + kk + slot] = (unsigned int)j; }
}

}  // namespace

int emb_plan_build(const EmbPlanSource& src, int B, int T, int W, long long Vw, long long nb, const EmbPlanArrays& out, long long totals_host[4]) {
  Engine& e = engine();
  hipStream_t s = e.stream;
  const long long P = (long long)B * (T + 1);
  const long long Vp = Vw * W;
  GOCTR_CHECK(Vp < (1ll << 31) && P < (1ll << 31), "embedding plan: vocabulary / batch too large for 32-bit keys");
  const unsigned int sentinel = (unsigned int)Vp;
  unsigned int bits = 1;
  while ((1ull << bits) <= (unsigned long long)Vp) ++bits;
  // batches per sort: as many as fit the 32-bit key next to the row index and the temporaries' budget (20 B per key + the sort's
  // own storage; EMB_PLAN_TMP_MB)
  const long long by_bits = 1ll << (32 - bits);
  const long long by_mem = std::max<long long>(1, (EMB_PLAN_TMP_MB << 20) / (32 * P));
  const long long chunk = std::max<long long>(1, std::min<long long>(std::min(by_bits, by_mem), nb));
  const long long nmax = chunk * P;
  GOCTR_CHECK(nmax < (1ll << 32), "embedding plan: chunk of %lld pairs does not fit 32-bit ranks", nmax);
  DevBuf<unsigned int> key_in, key_out, val_in, val_out, flag, rank, npairs, tiles;
  DevBuf<unsigned long long> scan_total;
  DevBuf<long long> totals;
  DevBuf<char> temp;
  if (key_in.alloc((size_t)nmax, false) || key_out.alloc((size_t)nmax, false) || val_in.alloc((size_t)nmax, false) ||
      val_out.alloc((size_t)nmax, false) || flag.alloc((size_t)nmax, false) || rank.alloc((size_t)nmax, false) || npairs.alloc((size_t)chunk) ||
      scan_total.alloc(1) || totals.alloc(4)) return -1;
  unsigned int kbits = 0;
  while ((1ll << kbits) < chunk) ++kbits;
  size_t temp_bytes = 0;
  GOCTR_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, key_in.p, key_out.p, val_in.p, val_out.p, (size_t)nmax, 0u, bits + kbits, s));
  if (temp.alloc(std::max<size_t>(temp_bytes, 16), false)) return -1;
  GOCTR_HIP(hipMemsetAsync(out.pair_off, 0, sizeof(long long), s));
  GOCTR_HIP(hipMemsetAsync(out.slot_base, 0, sizeof(long long), s));
  for (long long k0 = 0; k0 < nb; k0 += chunk) {
    const long long kc = std::min(chunk, nb - k0), n = kc * P;
    const dim3 g((unsigned)cdiv(n, 256));
    const KeysArgs ka{src.ub_ids, src.item_ids, src.rows, B, T, src.V, W, Vw, k0, P, bits, sentinel};
    GOCTR_HIP(hipMemsetAsync(npairs.p, 0, sizeof(unsigned int) * (size_t)kc, s));
    hipLaunchKernelGGL(emb_plan_keys_kernel, g, dim3(256), 0, s, ka, n, key_in.p, val_in.p);
    GOCTR_HIP(hipGetLastError());
    size_t tb = temp_bytes;
    GOCTR_HIP(rocprim::radix_sort_pairs(temp.p, tb, key_in.p, key_out.p, val_in.p, val_out.p, (size_t)n, 0u, bits + kbits, s));
    hipLaunchKernelGGL(emb_plan_heads_kernel, g, dim3(256), 0, s, key_out.p, n, P, bits, sentinel, flag.p, npairs.p);
    GOCTR_HIP(hipGetLastError());
    if (exclusive_scan(flag.p, n, rank.p, tiles, scan_total.p)) return -1;
    hipLaunchKernelGGL(emb_plan_offsets_kernel, dim3(1), dim3(1), 0, s, rank.p, flag.p, npairs.p, P, kc, k0, out.pair_off, out.slot_base, out.slot_off,
                       totals.p);
    const FillArgs fa{key_out.p, val_out.p, flag.p, rank.p, n, P, k0, bits, sentinel, W, Vw, out.pair_off, out.slot_base,
                      out.pair, out.pslot, out.pid, out.slot_id, out.slot_off};
    hipLaunchKernelGGL(emb_plan_fill_kernel, g, dim3(256), 0, s, fa);
    GOCTR_HIP(hipGetLastError());
  }
  return totals.download(totals_host, 4);       // (the one synchronisation of the build; the temporaries are released behind it)
}

}  // namespace goctr
