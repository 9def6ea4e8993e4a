// Corpus load + dictionary build on the device.
//
// Reference: word2vec.Train's prelude — memory.New / Corpus.Load (corpus/memory/memory.go:36-102) feeds every
// word of the ItemSeqGenerator channel (recommend/rcmd.go:539; the words are decimal item ids,
// example/movielens/feature.go:78) through dictionary.Add (dictionary.go:70-81): a word's id is its rank by FIRST
// APPEARANCE, cfs[id] its count; IndexedDoc (memory.go:53-62) drops the ids the MaxCount / MinCount filters hit
// (cpsutil.go:58-78).  In Go this is a string-keyed map insert per word on one goroutine.
//
// Here the tokens are int64 and stay in HBM.  The build is HBM/atomic-bound integer work, no sort:
//   1. dict_insert : open-addressing table (2x words, linear probing); per word one CAS on the key, an atomicMin on
//                    the slot's first position, an atomicAdd on its count; the word remembers its slot.
//   2. dict_flag   : flag[first position of every occupied slot] = 1
//   3. scan        : exclusive prefix sum of flag  ->  rank of each first position  =  the reference's id
//   4. dict_assign : id2key[id], cfs[id], slot -> id
//   5. doc_index   : idoc[pos] = id(slot_of[pos]);  keep flag from the two count filters
//   6. scan + doc_compact : IndexedDoc
// The result does not depend on the order the atomics land in: ids are a function of first positions only.
#include "corpus.h"
#include "scan.h"

#include <climits>
#include <memory>

#include "../../include/goctr.h"

using namespace goctr;

namespace {

constexpr long long KEY_EMPTY = (long long)0x8000000000000000ULL;   // INT64_MIN is not a valid token

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {   // splitmix64 finaliser
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27; x *= 0x94D049BB133111EBULL;
  x ^= x >> 31;
  return x;
}

__global__ void dict_clear_kernel(long long* tkey, unsigned long long* tfirst, unsigned long long* tcnt, long long slots) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < slots) { tkey[i] = KEY_EMPTY; tfirst[i] = ~0ULL; tcnt[i] = 0; }
}

__global__ void dict_insert_kernel(const long long* keys, long long n, long long* tkey, unsigned long long* tfirst,
                                   unsigned long long* tcnt, unsigned long long mask, unsigned int* slot_of) {
  const long long pos = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pos >= n) return;               // (only the tail wavefront is partial; the ballots below see its active lanes)
  const long long key = keys[pos];
  unsigned long long h = mix64((unsigned long long)key) & mask;
  for (;;) {
    long long cur = tkey[h];
    if (cur == KEY_EMPTY) cur = (long long)atomicCAS((unsigned long long*)&tkey[h], (unsigned long long)KEY_EMPTY, (unsigned long long)key);
    if (cur == KEY_EMPTY || cur == key) break;
    h = (h + 1) & mask;
  }
  slot_of[pos] = (unsigned int)h;
  // One atomic pair per DISTINCT slot of the wavefront instead of one per word: item popularity is Zipfian, so the few
  // hot slots would otherwise serialise hundreds of thousands of atomics on one L2 line.  Lanes are in position order,
  // so the lowest lane of a group holds the group's first position.
  unsigned long long todo = __ballot(1);
  const int lane = threadIdx.x & 63;
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned long long lh = __shfl(h, leader, 64);
    const unsigned long long same = __ballot(h == lh) & todo;
    if (lane == leader) {
      if (tfirst[h] > (unsigned long long)pos) atomicMin(&tfirst[h], (unsigned long long)pos);   // monotone: a stale read only costs an extra atomic
      atomicAdd(&tcnt[h], (unsigned long long)__popcll(same));
    }
    todo &= ~same;
  }
}

__global__ void dict_flag_kernel(const long long* tkey, const unsigned long long* tfirst, long long slots, unsigned int* flag) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < slots && tkey[i] != KEY_EMPTY) flag[tfirst[i]] = 1u;
}

__global__ void dict_assign_kernel(const long long* tkey, const unsigned long long* tfirst, const unsigned long long* tcnt,
                                   long long slots, const unsigned int* rank, unsigned int* slot_id, long long* id2key,
                                   long long* cfs) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= slots || tkey[i] == KEY_EMPTY) return;
  const unsigned int id = rank[tfirst[i]];
  slot_id[i] = id;
  id2key[id] = tkey[i];
  cfs[id] = (long long)tcnt[i];
}

// idoc + the IndexedDoc filter (cpsutil.go:58-78: MaxCount drops 0 < v && v < freq, MinCount drops 0 <= v && freq < v)
__global__ void doc_index_kernel(const unsigned int* slot_of, long long n, const unsigned int* slot_id, const long long* cfs,
                                 long long min_count, long long max_count, int* idoc, unsigned int* keep) {
  const long long pos = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pos >= n) return;
  const unsigned int id = slot_id[slot_of[pos]];
  const long long f = cfs[id];
  const bool drop = (0 < max_count && max_count < f) || (0 <= min_count && f < min_count);
  idoc[pos] = (int)id;
  keep[pos] = drop ? 0u : 1u;
}

__global__ void doc_compact_kernel(const int* idoc, const unsigned int* keep, const unsigned int* off, long long n, int* indexed) {
  const long long pos = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pos < n && keep[pos]) indexed[off[pos]] = idoc[pos];
}

}  // namespace

extern "C" {

int goctr_corpus_create(int64_t capacity_words, goctr_corpus** out) {
  GOCTR_ENTER();
  GOCTR_CHECK(out && capacity_words > 0 && capacity_words < (1LL << 31), "goctr_corpus_create: capacity must be in 1 .. 2^31-1 words");
  std::unique_ptr<goctr_corpus> c(new goctr_corpus);
  c->capacity = capacity_words;
  if (c->keys.alloc((size_t)capacity_words, false)) return -1;
  *out = c.release();
  return 0;
}

void goctr_corpus_destroy(goctr_corpus* c) { delete c; }

int goctr_corpus_append(goctr_corpus* c, const int64_t* keys, int64_t n) {
  GOCTR_ENTER_H(c);
  GOCTR_CHECK(c && (keys || n == 0) && n >= 0, "goctr_corpus_append: bad arguments");
  std::lock_guard<std::mutex> lk(c->mu);
  GOCTR_CHECK(c->n_words + n <= c->capacity, "goctr_corpus_append: %lld + %lld words exceed the capacity %lld",
              (long long)c->n_words, (long long)n, (long long)c->capacity);
  for (int64_t i = 0; i < n; ++i) GOCTR_CHECK(keys[i] != INT64_MIN, "goctr_corpus_append: token INT64_MIN is reserved");
  if (n && c->keys.upload(reinterpret_cast<const long long*>(keys), (size_t)n, (size_t)c->n_words)) return -1;
  c->n_words += n;
  c->built = false;
  return 0;
}

int goctr_corpus_build(goctr_corpus* c, int64_t min_count, int64_t max_count) {
  GOCTR_ENTER_H(c);
  GOCTR_CHECK(c, "goctr_corpus_build: null corpus");
  std::lock_guard<std::mutex> lk(c->mu);
  GOCTR_CHECK(c->n_words > 0, "goctr_corpus_build: empty corpus");
  const long long n = c->n_words;
  long long slots = 1024;
  while (slots < 2 * n) slots <<= 1;
  hipStream_t s = engine().stream;
  DevBuf<long long> tkey;
  DevBuf<unsigned long long> tfirst, tcnt, total;
  DevBuf<unsigned int> slot_of, flag, rank, slot_id, tiles;
  if (tkey.alloc((size_t)slots, false) || tfirst.alloc((size_t)slots, false) || tcnt.alloc((size_t)slots, false) ||
      total.alloc(1) || slot_of.alloc((size_t)n, false) || flag.alloc((size_t)n) || rank.alloc((size_t)n, false) ||
      slot_id.alloc((size_t)slots, false))
    return -1;
  const dim3 gs((unsigned)cdiv(slots, 256)), gn((unsigned)cdiv(n, 256)), b(256);
  hipLaunchKernelGGL(dict_clear_kernel, gs, b, 0, s, tkey.p, tfirst.p, tcnt.p, slots);
  hipLaunchKernelGGL(dict_insert_kernel, gn, b, 0, s, c->keys.p, n, tkey.p, tfirst.p, tcnt.p, (unsigned long long)(slots - 1), slot_of.p);
  hipLaunchKernelGGL(dict_flag_kernel, gs, b, 0, s, tkey.p, tfirst.p, slots, flag.p);
  GOCTR_HIP(hipGetLastError());
  if (exclusive_scan(flag.p, n, rank.p, tiles, total.p)) return -1;
  unsigned long long V = 0;
  if (total.download(&V, 1)) return -1;
  GOCTR_CHECK(V > 0 && V < (1ULL << 31), "goctr_corpus_build: dictionary size %llu out of range", V);
  c->V = (int64_t)V;
  if (c->id2key.alloc((size_t)V, false) || c->cfs.alloc((size_t)V, false) || c->idoc.alloc((size_t)n, false)) return -1;
  hipLaunchKernelGGL(dict_assign_kernel, gs, b, 0, s, tkey.p, tfirst.p, tcnt.p, slots, rank.p, slot_id.p, c->id2key.p, c->cfs.p);
  // flag / rank are reused as the keep flags and their offsets
  hipLaunchKernelGGL(doc_index_kernel, gn, b, 0, s, slot_of.p, n, slot_id.p, c->cfs.p, (long long)min_count, (long long)max_count,
                     c->idoc.p, flag.p);
  GOCTR_HIP(hipGetLastError());
  if (exclusive_scan(flag.p, n, rank.p, tiles, total.p)) return -1;
  unsigned long long kept = 0;
  if (total.download(&kept, 1)) return -1;
  c->n_indexed = (int64_t)kept;
  if (c->indexed.alloc((size_t)kept, false)) return -1;
  hipLaunchKernelGGL(doc_compact_kernel, gn, b, 0, s, c->idoc.p, flag.p, rank.p, n, c->indexed.p);
  GOCTR_HIP(hipGetLastError());
  GOCTR_HIP(hipStreamSynchronize(s));   // the scratch buffers above go out of scope
  c->built = true;
  return 0;
}

int goctr_corpus_info(goctr_corpus* c, int64_t* n_words, int64_t* V, int64_t* n_indexed) {
  GOCTR_CHECK(c, "goctr_corpus_info: null corpus");
  std::lock_guard<std::mutex> lk(c->mu);
  GOCTR_CHECK(c->built, "goctr_corpus_info: call goctr_corpus_build first");
  if (n_words) *n_words = c->n_words;
  if (V) *V = c->V;
  if (n_indexed) *n_indexed = c->n_indexed;
  return 0;
}

int goctr_corpus_get_dictionary(goctr_corpus* c, int64_t* id2key, int64_t* cfs) {
  GOCTR_ENTER_H(c);
  GOCTR_CHECK(c, "goctr_corpus_get_dictionary: null corpus");
  std::lock_guard<std::mutex> lk(c->mu);
  GOCTR_CHECK(c->built, "goctr_corpus_get_dictionary: call goctr_corpus_build first");
  if (id2key && c->id2key.download(reinterpret_cast<long long*>(id2key), (size_t)c->V)) return -1;
  if (cfs && c->cfs.download(reinterpret_cast<long long*>(cfs), (size_t)c->V)) return -1;
  return 0;
}

int goctr_corpus_get_doc(goctr_corpus* c, int32_t* idoc, int32_t* indexed) {
  GOCTR_ENTER_H(c);
  GOCTR_CHECK(c, "goctr_corpus_get_doc: null corpus");
  std::lock_guard<std::mutex> lk(c->mu);
  GOCTR_CHECK(c->built, "goctr_corpus_get_doc: call goctr_corpus_build first");
  if (idoc && c->idoc.download(idoc, (size_t)c->n_words)) return -1;
  if (indexed && c->n_indexed && c->indexed.download(indexed, (size_t)c->n_indexed)) return -1;
  return 0;
}

}  // extern "C"
