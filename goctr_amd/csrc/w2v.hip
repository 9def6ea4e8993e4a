// w2v.hip -- item2vec engine (float64, like the reference) + its C-ABI.
//
// Replaces embedding.TrainEmbedding (feature/embedding/wordemb.go:9-32, reference = auxten/go-ctr) ->
// word2vec.Train / train / trainPerThread / observe (model/word2vec/word2vec.go:90-243),
// skipGram.trainOne (model/word2vec/model.go:48-78), hierarchicalSoftmax.optim and
// negativeSampling.optim (model/word2vec/optimizer.go:52-129), the Huffman tree
// (corpus/dictionary/huffman.go:23-57, node/node.go:26-43), the sigmoid table (sigmoid_table.go) and the
// LCG (modelutil/modelutil.go:21-29).
//
// Two execution modes:
//   deterministic  ONE wavefront walks the doc in order; lane d owns embedding lane d; the dot product
//                  is summed in the reference's j = 0..dim-1 order (v_readlane broadcast) so every float64
//                  is bit-identical to a single-goroutine run of the reference algorithm.
//   hogwild        `streams` lane-groups (dim rounded up to a power of two lanes each) walk contiguous
//                  pieces of the doc concurrently and update the shared vectors without synchronisation:
//                  the reference's goroutine scheme (word2vec.go:151-175) with ~10^4 "goroutines".  The doc
//                  is cut into `slices` (IndexPerThread, the reference: runtime.NumCPU() of them) and every
//                  slice is shared by streams / slices workers: a worker's windows reach into its
//                  neighbours' pieces and are clipped only at the SLICE ends (quirk Q18), so the number of
//                  clipped windows is the reference's whatever parallelism the GPU needs.
//                  The learning-rate observer is replaced by a per-stream estimate of the global word
//                  count (no per-word channel send / atomic).
#include <algorithm>
#include <cmath>
#include <memory>
#include <numeric>

#include <chrono>
#include <thread>

#include "common.h"
#include "huffman.h"
#include "corpus.h"

using namespace goctr;

namespace {

struct W2vDev {
  int dim, window, optimizer, neg, model;
  double init_lr, min_lr;
  long long update_lr_batch;
  long long V;
  double* param; double* aux;
  const long long* path_off; const int* path_nodes; const unsigned char* path_codes;
  const double* sigtab;
  const int* doc; const unsigned char* keep;  // keep may be null
  long long n_words, corpus_len;
  double* lr;              // in/out (deterministic) / in (hogwild)
  unsigned long long* lcg; // shared LCG state (deterministic)
  long long* trained;      // observer counter
  // hogwild launches: segment `seg` of `nseg` equal parts of every stream's piece (data-parallel passes exchange parameter
  // deltas between segments; 0 of 1 = the whole pass), the ranks sharing the pass (the observer estimate counts THEIR words
  // too: the reference's schedule runs on the global trained-word count, word2vec.go:223-233) and this rank's first stream
  // number (stream seeds differ between ranks)
  int seg, nseg;
  long long est_scale;
  long long seed_base;
};

__device__ __forceinline__ int lcg_next(unsigned long long& next, int value) {
  next = next * 25214903917ULL + 11ULL;  // modelutil.go:26-29
  return (int)(next % (unsigned long long)value);
}

__device__ __forceinline__ double sig_lookup(const double* tab, double x) {
  return tab[(int)((x + 6.0) * (1000.0 / 6.0 / 2.0))];  // sigmoid_table.go:43-45
}

// wave-uniform sequential sum of the first `dim` lanes' values, j = 0..dim-1 (bit-exact vs the Go loop)
__device__ __forceinline__ double seq_sum(double v, int dim) {
  double s = 0;
  for (int j = 0; j < dim; ++j) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
    s += __hiloint2double(hi, lo);
  }
  return s;
}

// Hogwild's shared vectors are read and updated by workgroups on all 8 XCDs, whose L2s are not coherent with each other
// (MI355X_MICROARCH.md "Correctness boundaries"): a plain load keeps hitting the XCD's own stale line for as long as the
// 2.7 MB of parameters stay L2-resident (= the whole pass), and a plain read-modify-write store loses every update that
// raced with it.  So: device-scope loads (sc1) and device-scope atomic adds -- an update is never lost, and a reader sees
// what the other XCDs have contributed so far, which is what the reference's goroutines get from a coherent CPU cache.
__device__ __forceinline__ double hog_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hog_add(double* p, double v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the deterministic single-wavefront pass keeps plain accesses (one wavefront, program order: bit-exact vs the oracle)
template <bool HOG> __device__ __forceinline__ double w2v_ld(const double* p) { return HOG ? hog_load(p) : *p; }
template <bool HOG> __device__ __forceinline__ void w2v_upd(double* p, double old, double delta) {
  if (HOG) hog_add(p, delta); else *p = old + delta;
}

// One optimizer call (optimizer.go:52-91 / :107-129) for the lane that owns component l of the vectors:
// ctx = that component of the input vector, tmp accumulates the component of the input's update.
// `sum` is the inner-product reduction (sequential for the deterministic mode, butterfly for Hogwild).
// OPT: -1 = a.optimizer decides at run time; 0 / 1 = hierarchical softmax / negative sampling fixed at compile time
template <bool HOG, int OPT = -1, class Sum>
__device__ __forceinline__ void w2v_optim(const W2vDev& a, const double* tab, int id, double lr, double ctx, double& tmp,
                                          unsigned long long& next, bool act, int l, Sum sum) {
  const int dim = a.dim;
  if (OPT < 0 ? a.optimizer == 0 : OPT == 0) {
    for (long long i = a.path_off[id]; i < a.path_off[id + 1]; ++i) {
      double* pvp = a.aux + (long long)a.path_nodes[i] * dim + l;
      const double pv = act ? w2v_ld<HOG>(pvp) : 0.0;
      const double inner = sum(ctx * pv);
      if (inner <= -6.0 || inner >= 6.0) break;  // quirk Q13: `return`
      const double g = (1.0 - (double)a.path_codes[i] - sig_lookup(tab, inner)) * lr;
      tmp += g * pv;
      if (act) w2v_upd<HOG>(pvp, pv, g * ctx);
    }
  } else {
    for (int n = -1; n < a.neg; ++n) {
      int label, picked;
      if (n == -1) { label = 1; picked = id; }
      else {
        label = 0;
        picked = lcg_next(next, (int)a.V);
        if (id == picked) continue;
      }
      double* rp = a.aux + (long long)picked * dim + l;
      const double rnd = act ? w2v_ld<HOG>(rp) : 0.0;
      const double inner = sum(rnd * ctx);
      double g;
      if (inner <= -6.0) g = ((double)(label - 0)) * lr;
      else if (inner >= 6.0) g = ((double)(label - 1)) * lr;
      else g = ((double)label - sig_lookup(tab, inner)) * lr;
      tmp += g * rnd;
      if (act) w2v_upd<HOG>(rp, rnd, g * ctx);
    }
  }
}

// cbow.trainOne (model.go:96-148): aggregate the window's vectors, one optimizer call on the aggregate, add its
// update to every window vector.  The window shrink is drawn twice (once in the aggregate pass, once in the update
// pass — `dowith` calls NextRandom each time), so the two passes may cover different windows.
template <bool HOG, int OPT = -1, class Sum>
__device__ __forceinline__ void w2v_cbow_one(const W2vDev& a, const double* tab, const int* doc, long long cmin, long long cmax,
                                             long long pos, double lr, unsigned long long& next, bool act, int l, Sum sum) {
  const int dim = a.dim, win = a.window;
  double agg = 0.0, tmp = 0.0;
  int del = lcg_next(next, win);
  for (int w = del; w < win * 2 + 1 - del; ++w) {
    if (w == win) continue;
    const long long c = pos - win + w;
    if (c < cmin || c >= cmax) continue;
    if (act) agg += w2v_ld<HOG>(a.param + (long long)doc[c] * dim + l);
  }
  w2v_optim<HOG, OPT>(a, tab, doc[pos], lr, agg, tmp, next, act, l, sum);
  del = lcg_next(next, win);
  for (int w = del; w < win * 2 + 1 - del; ++w) {
    if (w == win) continue;
    const long long c = pos - win + w;
    if (c < cmin || c >= cmax) continue;
    if (act) {   // a word twice in the window gets the update twice
      double* wp = a.param + (long long)doc[c] * dim + l;
      w2v_upd<HOG>(wp, HOG ? 0.0 : *wp, tmp);
    }
  }
}

// ---- deterministic single-stream pass: one block of 64 threads
__global__ __launch_bounds__(64) void w2v_deterministic_kernel(W2vDev a) {
  __shared__ double tab[1000];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1000; i += 64) tab[i] = a.sigtab[i];
  __syncthreads();
  const int dim = a.dim, win = a.window;
  const bool act = lane < dim;
  unsigned long long next = *a.lcg;
  double lr = *a.lr;
  long long cnt = *a.trained;
  for (long long pos = 0; pos < a.n_words; ++pos) {
    const int id = a.doc[pos];
    if (a.model == 1) {
      if (!a.keep || a.keep[pos]) {
        w2v_cbow_one<false>(a, tab, a.doc, 0, a.n_words, pos, lr, next, act, lane, [&](double v) { return seq_sum(v, dim); });
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
    } else if (!a.keep || a.keep[pos]) {
      const int del = lcg_next(next, win);  // model.go:59
      for (int w = del; w < win * 2 + 1 - del; ++w) {
        if (w == win) continue;
        const long long c = pos - win + w;
        if (c < 0 || c >= a.n_words) continue;
        const int ctxid = a.doc[c];
        double* ctxp = a.param + (long long)ctxid * dim + lane;
        double ctx = act ? *ctxp : 0.0, tmp = 0.0;
        if (a.optimizer == 0) {  // hierarchical softmax, optimizer.go:107-129
          for (long long i = a.path_off[id]; i < a.path_off[id + 1]; ++i) {
            double* pvp = a.aux + (long long)a.path_nodes[i] * dim + lane;
            double pv = act ? *pvp : 0.0;
            const double inner = seq_sum(ctx * pv, dim);
            if (inner <= -6.0 || inner >= 6.0) break;  // quirk Q13: `return`
            const double g = (1.0 - (double)a.path_codes[i] - sig_lookup(tab, inner)) * lr;
            tmp += g * pv;
            pv += g * ctx;
            if (act) *pvp = pv;
          }
        } else {  // negative sampling, optimizer.go:52-91
          for (int n = -1; n < a.neg; ++n) {
            int label, picked;
            if (n == -1) { label = 1; picked = id; }
            else {
              label = 0;
              picked = lcg_next(next, (int)a.V);
              if (id == picked) continue;
            }
            double* rp = a.aux + (long long)picked * dim + lane;
            double rnd = act ? *rp : 0.0;
            const double inner = seq_sum(rnd * ctx, dim);
            double g;
            if (inner <= -6.0) g = ((double)(label - 0)) * lr;
            else if (inner >= 6.0) g = ((double)(label - 1)) * lr;
            else g = ((double)label - sig_lookup(tab, inner)) * lr;
            tmp += g * rnd;
            rnd += g * ctx;
            if (act) *rp = rnd;
          }
        }
        ctx += tmp;  // model.go:74-76
        if (act) *ctxp = ctx;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
    }
    ++cnt;  // observe(): word2vec.go:223-233
    if (cnt % a.update_lr_batch == 0) {
      if (lr < a.min_lr) lr = a.min_lr;
      else lr = a.init_lr * (1.0 - (double)cnt / (double)a.corpus_len);
    }
  }
  if (lane == 0) { *a.lcg = next; *a.lr = lr; *a.trained = cnt; }
}

// Sum over the GS lanes of a lane group, the same bits in every lane (the lanes branch on it).  Steps inside a 16-lane DPP
// row are VALU moves with a lane pattern (two per double) -- round 4: they were ds_bpermute pairs (__shfl_xor), eight LDS-crossbar
// round trips per inner product, which is what the Hogwild walk spent its time in once the node traffic was cut.  Pairings:
// quad_perm xor 1 / xor 2, then within 8 lanes the mirror (i <-> 7 - i), within 16 the rotation by 8 (= xor 8); every step adds
// the same two values in both partner lanes (commutative: identical bits), so the group agrees on the result.
template <int CTRL>
__device__ __forceinline__ double dpp_add64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
template <int GS>
__device__ __forceinline__ double group_sum64(double v) {
  if (GS >= 16) v = dpp_add64<0x128>(v);             // row_ror:8   lane i + lane (i + 8) % 16
  if (GS >= 8) {
    if (GS >= 16) v = dpp_add64<0x124>(v);           // row_ror:4   -> all lanes = i (mod 4)
    else v = dpp_add64<0x141>(v);                    // row_half_mirror (GS = 8): lane i + lane 7 - i
  }
  v = dpp_add64<0x4E>(v);                            // quad_perm [2,3,0,1]
  v = dpp_add64<0xB1>(v);                            // quad_perm [1,0,3,2]
#pragma unroll
  for (int o = 16; o < GS; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- hogwild: one lane-group (GS lanes) per stream = per contiguous piece [slice_idx[g], slice_idx[g+1]) of the doc;
// window clipping is against the SLICE (IndexPerThread, modelutil.go:32-41; quirk Q18) the piece belongs to,
// [clip_lo[g], clip_hi[g]).
//
// Hot rows live in LDS.  Every update walks the Huffman path from the root, and item popularity is
// Zipfian: the few hundred heaviest inner nodes and most frequent words take most of the read-modify-writes.  As
// device-scope atomics on a handful of cache lines those serialise (measured: 4.3 M words/s with every access at device
// scope); as plain stores they are lost (and stale: the XCDs' L2s are not coherent), which is what cost the first
// version 5 % of HS loss against the oracle's 16-thread run.  So each workgroup keeps a private copy of the HOT_ROWS
// heaviest node vectors and most frequent word vectors in LDS -- read and updated there by its 1024 / GS lane groups,
// racing like the reference's goroutines do -- and every `merge_every` words folds its accumulated delta into the global
// row (device-scope atomic add) and takes the other workgroups' contributions back (device-scope load).  The delta
// enters scaled by 1 / workgroups, i.e. the replicas of a hot row are AVERAGED: summing them was measured to diverge
// (HS loss 18 .. 680 instead of 0.56) -- a row that takes a share p of all updates sees  rate x p x latency  of them
// concurrently, and SGD on one vector is only stable up to ~100 stale updates at lr 0.025; the root (p = 1) would need
// 20 ns visibility.  Averaging costs the hot rows nothing they need (they see 10^5 .. 10^7 updates each) and measured
// 0.565 vs the oracle's 0.559 (16 threads) at 10^7 words; cold rows -- few updates each, none to waste -- take the
// exact path: device-scope load + atomic add straight to memory.
constexpr int HOG_THREADS = 1024;
// doubles per cached table.  Round 4: the hot copies' BASE values (what a copy held at its last merge: delta = copy - base) moved
// from LDS to a workgroup-private strip of global memory -- they are touched only at the merges, 64 KB per workgroup read and
// written once per 16 positions = 128 B per word of plain cached traffic -- so the same 72 KB of LDS (2 tables x 32 KB + the 8 KB
// sigmoid table; two workgroups per CU) now hold TWICE the rows: 256 node vectors + 256 word vectors at dim 16.  With Zipfian
// counts a Huffman path's nodes halve in weight per level, so 256 cached nodes cover one more level of every walk than 128
// (cold read-modify-writes per pair 3.05 -> ~2.0 estimated at V = 10 681).
constexpr int HOG_HOT_DOUBLES = 4096;
// doubles per cached table of the kernel variant with WPS wavefronts per SIMD: at 4 (one 1024-thread workgroup per CU, 128
// registers per lane) the workgroup has the CU's LDS to itself and caches twice the rows
constexpr int hog_hot_doubles(int wps) { return wps <= 4 ? 2 * HOG_HOT_DOUBLES : HOG_HOT_DOUBLES; }

struct HogHot {
  const int* word_slot;     // [V] slot of a hot word in the LDS cache or -1
  const int* word_id;       // [n_words_hot] slot -> word
  int n_nodes, n_words;     // rows cached of aux (the LAST n_nodes rows = the heaviest Huffman nodes) and of param
  long long node0;          // first cached aux row
  double* base;             // [workgroups][2][hog_hot_doubles(WPS)] base values of the hot copies (global, private per workgroup)
  int merge_every;          // words per lane group between merges
  double merge_scale;       // a workgroup's delta enters the global row times this (1 / workgroups: the replicas are averaged)
  long long max_len;        // longest piece (uniform loop bound: every thread meets every barrier)
};

#ifndef HOG_PF_N
#define HOG_PF_N 4
#endif
constexpr int HOG_PF = HOG_PF_N;   // node vectors of a Huffman path in flight per lane group
#ifndef HOG_WAVES_PER_SIMD
#define HOG_WAVES_PER_SIMD 8
#endif

// MODEL (0 skip-gram, 1 cbow) and OPT (0 hierarchical softmax, 1 negative sampling) are compile-time: one kernel holding all
// four combinations needs 103 registers, and the 64 that let two 1024-thread workgroups share a CU (8 wavefronts per SIMD
// instead of 4 -- the walk is a chain of dependent loads, more lane groups in flight is what it wants) are then 38 spilled
template <int GS, int MODEL, int OPT>
__global__ __launch_bounds__(HOG_THREADS, HOG_WAVES_PER_SIMD) void w2v_hogwild_kernel(W2vDev a, int streams, const long long* slice_idx, const long long* clip_lo,
                                                                  const long long* clip_hi, HogHot hot) {
  constexpr int HOT = HOG_HOT_DOUBLES;
  __shared__ double tab[1000];
  __shared__ double locN[HOT], locW[HOT];
  double* const baseN = hot.base + (size_t)blockIdx.x * 2 * HOT;      // (only this workgroup reads or writes its strip)
  double* const baseW = baseN + HOT;
  const int dim = a.dim, win = a.window;
  for (int i = threadIdx.x; i < 1000; i += HOG_THREADS) tab[i] = a.sigtab[i];
  // fill the caches (row stride GS doubles)
  for (int i = threadIdx.x; i < hot.n_nodes * GS; i += HOG_THREADS) {
    const int r = i / GS, c = i % GS;
    const double v = c < dim ? hog_load(a.aux + (hot.node0 + r) * dim + c) : 0.0;
    locN[i] = v; baseN[i] = v;
  }
  for (int i = threadIdx.x; i < hot.n_words * GS; i += HOG_THREADS) {
    const int r = i / GS, c = i % GS;
    const double v = c < dim ? hog_load(a.param + (long long)hot.word_id[r] * dim + c) : 0.0;
    locW[i] = v; baseW[i] = v;
  }
  __syncthreads();
  // add this workgroup's delta to the global rows, take the others' contributions back
  auto merge = [&]() {
    __syncthreads();
    for (int i = threadIdx.x; i < hot.n_nodes * GS; i += HOG_THREADS) {
      const int r = i / GS, c = i % GS;
      if (c < dim) {
        double* gp = a.aux + (hot.node0 + r) * dim + c;
        const double d = (locN[i] - baseN[i]) * hot.merge_scale;
        if (d != 0.0) hog_add(gp, d);
        const double v = hog_load(gp);
        locN[i] = v; baseN[i] = v;
      }
    }
    for (int i = threadIdx.x; i < hot.n_words * GS; i += HOG_THREADS) {
      const int r = i / GS, c = i % GS;
      if (c < dim) {
        double* gp = a.param + (long long)hot.word_id[r] * dim + c;
        const double d = (locW[i] - baseW[i]) * hot.merge_scale;
        if (d != 0.0) hog_add(gp, d);
        const double v = hog_load(gp);
        locW[i] = v; baseW[i] = v;
      }
    }
    __syncthreads();
  };
  constexpr int GPB = HOG_THREADS / GS;  // groups per block
  const int g = blockIdx.x * GPB + threadIdx.x / GS;
  const int l = threadIdx.x % GS;
  const bool act = l < dim && g < streams;
  const int gs = g < streams ? g : streams - 1;
  const long long lo0 = slice_idx[gs], len0 = g < streams ? slice_idx[gs + 1] - lo0 : 0;  // idle groups run 0 words
  const long long pb = len0 * a.seg / a.nseg;                      // this launch's part of the piece (whole piece: 0 of 1)
  const long long lo = lo0 + pb, hi = lo0 + len0 * (a.seg + 1) / a.nseg;
  // per-stream LCG; stream 0 of rank 0 starts its first segment from the reference's seed (modelutil.go:21-24)
  unsigned long long next = 1ULL + 0x9E3779B97F4A7C15ULL * ((unsigned long long)(a.seed_base + g) + (unsigned long long)a.seg * 0x100000000ULL);
  const double lr0 = *a.lr;
  double lr = lr0;
  long long est = pb * streams * a.est_scale, at = est / a.update_lr_batch * a.update_lr_batch;
  const int* doc = a.doc + lo;
  const long long len = hi - lo;
  const long long cmin = clip_lo[gs] - lo, cmax = clip_hi[gs] - lo;   // window positions allowed, relative to this piece
  // vector component l of a word / of an inner node (HS) -- LDS when hot, device-scope memory access otherwise
  auto word_slot = [&](int id) { return hot.n_words ? hot.word_slot[id] : -1; };
  auto ld_word = [&](int id, int slot) { return slot >= 0 ? locW[slot * GS + l] : hog_load(a.param + (long long)id * dim + l); };
  auto add_word = [&](int id, int slot, double v) {
    if (slot >= 0) locW[slot * GS + l] += v; else hog_add(a.param + (long long)id * dim + l, v);
  };
  auto ld_node = [&](int nd) {
    return nd >= hot.node0 ? locN[(nd - (int)hot.node0) * GS + l] : hog_load(a.aux + (long long)nd * dim + l);
  };
  auto add_node = [&](int nd, double v) {
    if (nd >= hot.node0) locN[(nd - (int)hot.node0) * GS + l] += v; else hog_add(a.aux + (long long)nd * dim + l, v);
  };
  for (long long pos = 0; pos < hot.max_len; ++pos) {
    if (pos < len) {
      const int id = doc[pos];
      if (MODEL == 1) {
        if (!a.keep || a.keep[lo + pos])
          w2v_cbow_one<true, OPT>(a, tab, doc, cmin, cmax, pos, lr, next, act, l, [&](double v) { return group_sum64<GS>(v); });
      } else if (!a.keep || a.keep[lo + pos]) {
        const int del = lcg_next(next, win);
        // every pair of this position walks the SAME Huffman path (the centre word's): its first GS nodes are fetched once
        // (32-bit path offsets: the host refuses trees with 2^31 path entries or more)
        const int hp0 = OPT == 0 ? (int)a.path_off[id] : 0, hp1 = OPT == 0 ? (int)a.path_off[id + 1] : 0;
        const int hn0 = hp1 - hp0 < GS ? hp1 - hp0 : GS;
        const int h_nd0 = l < hn0 ? a.path_nodes[hp0 + l] : 0;
        const int h_code0 = l < hn0 ? (int)a.path_codes[hp0 + l] : 0;
        // ... and the context vector of the NEXT pair is requested before the current pair's walk starts
        auto next_ctx = [&](int w_from, int& w_out, int& cid_out, int& cslot_out, double& v_out) {
          w_out = win * 2 + 1;
          for (int w = w_from; w < win * 2 + 1 - del; ++w) {
            if (w == win) continue;
            const long long c = pos - win + w;
            if (c < cmin || c >= cmax) continue;
            w_out = w; cid_out = doc[c]; cslot_out = word_slot(cid_out);
            v_out = act ? ld_word(cid_out, cslot_out) : 0.0;
            return;
          }
        };
        int w_n = 0, cid_n = 0, cslot_n = -1; double ctx_n = 0.0;
        next_ctx(del, w_n, cid_n, cslot_n, ctx_n);
        while (w_n < win * 2 + 1 - del) {
          const int cid = cid_n, cslot = cslot_n;
          double ctx = ctx_n, tmp = 0.0;
          {
            // (a repeated context word in the window must see the previous pair's update: then the row is re-read after it)
            int w2 = 0, cid2 = 0, cslot2 = -1; double v2 = 0.0;
            next_ctx(w_n + 1, w2, cid2, cslot2, v2);
            w_n = w2; cid_n = cid2; cslot_n = cslot2; ctx_n = v2;
          }
          if (OPT == 0) {
            // The path is known up front: its node ids and codes arrive with ONE coalesced load per GS nodes (lane k of the
            // group holds node k, handed round by shuffle), and the node vectors are requested HOG_PF nodes ahead -- a
            // device-scope load of a cold node takes microseconds, and with one node in flight the walk ran at one such
            // latency per node.  (The nodes of a path are distinct, so reading ahead skips no update of this walk.)
            const int p0 = hp0, p1 = hp1;
            const int gbase = (int)(threadIdx.x & 63) & ~(GS - 1);
            for (int c0 = p0; c0 < p1; c0 += GS) {
              const int n = p1 - c0 < GS ? p1 - c0 : GS;
              const int my_nd = c0 == p0 ? h_nd0 : (l < n ? a.path_nodes[c0 + l] : 0);
              const int my_code = c0 == p0 ? h_code0 : (l < n ? (int)a.path_codes[c0 + l] : 0);
              double pf[HOG_PF];
#pragma unroll
              for (int k = 0; k < HOG_PF; ++k) {
                const int ndk = __shfl(my_nd, gbase + (k < n ? k : 0), 64);
                pf[k] = (act && k < n) ? ld_node(ndk) : 0.0;
              }
              bool stop = false;
              for (int i = 0; i < n; ++i) {
                const int nd = __shfl(my_nd, gbase + i, 64);
                const int code = __shfl(my_code, gbase + i, 64);
                const double pv = pf[0];
#pragma unroll
                for (int k = 0; k + 1 < HOG_PF; ++k) pf[k] = pf[k + 1];
                {
                  const int ia = i + HOG_PF;
                  const int nda = __shfl(my_nd, gbase + (ia < n ? ia : 0), 64);
                  pf[HOG_PF - 1] = (act && ia < n) ? ld_node(nda) : 0.0;
                }
                const double inner = group_sum64<GS>(ctx * pv);
                if (inner <= -6.0 || inner >= 6.0) { stop = true; break; }
                const double gg = (1.0 - (double)code - sig_lookup(tab, inner)) * lr;
                tmp += gg * pv;
                if (act) add_node(nd, gg * ctx);          // pv += g * ctx (optimizer.go:125)
              }
              if (stop) break;
            }
          } else {
            for (int n = -1; n < a.neg; ++n) {
              int label, picked;
              if (n == -1) { label = 1; picked = id; }
              else {
                label = 0;
                picked = lcg_next(next, (int)a.V);
                if (id == picked) continue;
              }
              double* rp = a.aux + (long long)picked * dim + l;   // (negatives are uniform draws: no hot rows to cache)
              double rnd = act ? hog_load(rp) : 0.0;
              const double inner = group_sum64<GS>(rnd * ctx);
              double gg;
              if (inner <= -6.0) gg = ((double)(label - 0)) * lr;
              else if (inner >= 6.0) gg = ((double)(label - 1)) * lr;
              else gg = ((double)label - sig_lookup(tab, inner)) * lr;
              tmp += gg * rnd;
              if (act) hog_add(rp, gg * ctx);
            }
          }
          if (act) add_word(cid, cslot, tmp);            // ctx += tmp (model.go:74-76)
          if (w_n < win * 2 + 1 - del && cid_n == cid && act) ctx_n = ld_word(cid_n, cslot_n);   // same word again: re-read
        }
      }
      // observer estimate: all streams advance at the same rate => global count ~= positions so far * streams; the rate is
      // re-derived whenever that estimate passes a multiple `at` of update_lr_batch (word2vec.go:223-233).  Kept as a
      // running multiple: two 64-bit divisions per position were ~300 instructions and a dozen registers of this loop.
      est += streams * a.est_scale;
      if (est >= at + a.update_lr_batch) {
        do at += a.update_lr_batch; while (est >= at + a.update_lr_batch);
        if (lr < a.min_lr) lr = a.min_lr;
        else lr = a.init_lr * (1.0 - (double)at / (double)a.corpus_len);
      }
    }
    if ((pos + 1) % hot.merge_every == 0) merge();
  }
  merge();
  if (g == 0 && l == 0) *a.trained = a.n_words;
  if (g == streams - 1 && l == 0) *a.lr = lr;  // the lr the last words saw
}

// ---- hogwild, skip-gram + hierarchical softmax, NODE-MAJOR (round 4).  Every pair of a position walks the SAME Huffman path
// (the centre word's) with its own context vector; pair-major order -- the reference's, model.go:60-77, and the kernel above --
// reads and updates every node of the path once per PAIR: 2 x (window - shrink) ~ 6 device-scope loads and atomic adds per cold
// node and position.  Here the pairs of a position are walked JB at a time, node by node: a node vector is read ONCE per chunk,
// pair j + 1 sees pair j's update in a register (exactly what it would have read back: within a stream the arithmetic is the
// sequential one -- pair j at node i still sees the updates of pairs < j at node i, and its own context vector as it was when
// its walk began), and the chunk's summed update leaves with ONE atomic add.  A context word that occurs twice in a window
// starts a new chunk (its second walk must begin from the first's result).  Other streams' updates of a node arrive between
// chunks instead of between pairs: Hogwild's race window, a few hundred nanoseconds either way.
//
// CPL components per lane: a lane group is GS lanes holding GS x CPL >= dim components (component c = l + k GS in lane l), so
// a wavefront carries 64 / GS streams.  What is per PAIR AND NODE and the same in all lanes of a group -- the range test, the
// sigmoid lookup, the gradient scalar -- is paid once per group: at dim 16, 8 lanes x 2 components halve that share per stream
// and drop one reduction step, and 128 registers (WPS = 4: one workgroup per CU, which then also has the CU's LDS to itself)
// hold the 2 JB context / update vectors without spilling.
// (Round 5 flattened the position > chunk > node nest -- one lockstep iteration = one chunk of each stream's own position, lane
// efficiency 0.45 -> ~0.68: 13-18 % fewer vector instructions, the pass no shorter; profiles/r05_w2v_flat_ab.txt.  Removed.)
template <int GS, int CPL, int JB, int WPS, int PF>
__global__ __launch_bounds__(HOG_THREADS, WPS) void w2v_hogwild_nm_kernel(W2vDev a, int streams, const long long* slice_idx, const long long* clip_lo,
                                                                         const long long* clip_hi, HogHot hot) {
  constexpr int HOT = hog_hot_doubles(WPS);
  constexpr int RS = GS * CPL;                 // row stride of the LDS tables
  __shared__ double tab[1000];
  __shared__ double locN[HOT], locW[HOT];
  double* const baseN = hot.base + (size_t)blockIdx.x * 2 * HOT;      // (only this workgroup reads or writes its strip)
  double* const baseW = baseN + HOT;
  const int dim = a.dim, win = a.window;
  for (int i = threadIdx.x; i < 1000; i += HOG_THREADS) tab[i] = a.sigtab[i];
  for (int i = threadIdx.x; i < hot.n_nodes * RS; i += HOG_THREADS) {
    const int r = i / RS, c = i % RS;
    const double v = c < dim ? hog_load(a.aux + (hot.node0 + r) * dim + c) : 0.0;
    locN[i] = v; baseN[i] = v;
  }
  for (int i = threadIdx.x; i < hot.n_words * RS; i += HOG_THREADS) {
    const int r = i / RS, c = i % RS;
    const double v = c < dim ? hog_load(a.param + (long long)hot.word_id[r] * dim + c) : 0.0;
    locW[i] = v; baseW[i] = v;
  }
  __syncthreads();
  auto merge = [&]() {                         // (see w2v_hogwild_kernel)
    __syncthreads();
    for (int i = threadIdx.x; i < hot.n_nodes * RS; i += HOG_THREADS) {
      const int r = i / RS, c = i % RS;
      if (c < dim) {
        double* gp = a.aux + (hot.node0 + r) * dim + c;
        const double d = (locN[i] - baseN[i]) * hot.merge_scale;
        if (d != 0.0) hog_add(gp, d);
        const double v = hog_load(gp);
        locN[i] = v; baseN[i] = v;
      }
    }
    for (int i = threadIdx.x; i < hot.n_words * RS; i += HOG_THREADS) {
      const int r = i / RS, c = i % RS;
      if (c < dim) {
        double* gp = a.param + (long long)hot.word_id[r] * dim + c;
        const double d = (locW[i] - baseW[i]) * hot.merge_scale;
        if (d != 0.0) hog_add(gp, d);
        const double v = hog_load(gp);
        locW[i] = v; baseW[i] = v;
      }
    }
    __syncthreads();
  };
  constexpr int GPB = HOG_THREADS / GS;
  const int g = blockIdx.x * GPB + threadIdx.x / GS;
  const int l = threadIdx.x % GS;
  const int gs = g < streams ? g : streams - 1;
  const long long lo0 = slice_idx[gs], len0 = g < streams ? slice_idx[gs + 1] - lo0 : 0;   // (the host refuses pieces of 2^31 words or more)
  const long long pb = len0 * a.seg / a.nseg;                      // this launch's part of the piece (whole piece: 0 of 1)
  const long long lo = lo0 + pb;
  const int len = (int)(len0 * (a.seg + 1) / a.nseg - pb);
  bool actk[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) actk[k] = l + k * GS < dim && g < streams;
  // (stream 0 of rank 0, first segment: the reference's seed)
  unsigned long long next = 1ULL + 0x9E3779B97F4A7C15ULL * ((unsigned long long)(a.seed_base + g) + (unsigned long long)a.seg * 0x100000000ULL);
  double lr = *a.lr;
  long long est = pb * streams * a.est_scale, at = est / a.update_lr_batch * a.update_lr_batch;
  const int* doc = a.doc + lo;
  const unsigned char* keep = a.keep ? a.keep + lo : nullptr;
  const long long cmin = clip_lo[gs] - lo, cmax = clip_hi[gs] - lo;
  const int gbase = (int)(threadIdx.x & 63) & ~(GS - 1);
  auto ld_node = [&](int nd, bool on, double (&v)[CPL]) {
#pragma unroll
    for (int k = 0; k < CPL; ++k)
      v[k] = !(on && actk[k]) ? 0.0
             : (nd >= hot.node0 ? locN[(nd - (int)hot.node0) * RS + l + k * GS] : hog_load(a.aux + (long long)nd * dim + l + k * GS));
  };
  auto add_node = [&](int nd, const double (&v)[CPL]) {
#pragma unroll
    for (int k = 0; k < CPL; ++k)
      if (actk[k] && v[k] != 0.0) { if (nd >= hot.node0) locN[(nd - (int)hot.node0) * RS + l + k * GS] += v[k]; else hog_add(a.aux + (long long)nd * dim + l + k * GS, v[k]); }
  };
  for (long long pos = 0; pos < hot.max_len; ++pos) {
    if (pos < len) {
      if (!keep || keep[pos]) {
        const int id = doc[pos];
        const int del = lcg_next(next, win);
        // the path's node ids and codes arrive with ONE coalesced load per GS nodes (lane k holds node k, handed round by shuffle)
        const int hp0 = (int)a.path_off[id], hp1 = (int)a.path_off[id + 1];
        const int hn0 = hp1 - hp0 < GS ? hp1 - hp0 : GS;
        const int h_nd0 = l < hn0 ? a.path_nodes[hp0 + l] : 0;
        const int h_code0 = l < hn0 ? (int)a.path_codes[hp0 + l] : 0;
        const int wend = win * 2 + 1 - del;
        int w = del;
        while (w < wend) {
          // A chunk = the next (at most JB) context words in window order, up to the first repeated id (which opens the next chunk).
          // Opened in ROUNDS -- window offsets (bounds only), then all ids, then all slots, then all vectors: written pair by pair
          // (find, test, load the vector, next pair) every pair's three dependent loads were waited for before the next pair's
          // first was issued (vmcnt counts in order: the scan's doc[c] drains the vector loads in front of it) -- twelve serial
          // round trips per chunk where three do (profiles/r06_w2v_rounds.txt).
          int cid[JB]; double ctx[JB][CPL], tmp[JB][CPL];
          int cw[JB]; int ncand = 0;
          {
            int ws = w;
#pragma unroll
            for (int j = 0; j < JB; ++j) {
              cw[j] = wend;
              if (ncand == j) {
                for (; ws < wend; ++ws) {
                  if (ws == win) continue;
                  const long long c = pos - win + ws;
                  if (c < cmin || c >= cmax) continue;
                  break;
                }
                if (ws < wend) { cw[j] = ws; ++ws; ncand = j + 1; }
              }
            }
          }
          int fid[JB];
#pragma unroll
          for (int j = 0; j < JB; ++j) fid[j] = doc[pos - win + (j < ncand ? cw[j] : win)];     // (past the candidates: the centre word, unused)
          int nj = 0;
#pragma unroll
          for (int j = 0; j < JB; ++j) {
            cid[j] = -1;
#pragma unroll
            for (int k = 0; k < CPL; ++k) { ctx[j][k] = 0.0; tmp[j][k] = 0.0; }
            if (nj == j && j < ncand) {                          // (the chunk is still open)
              bool dup = false;
#pragma unroll
              for (int k = 0; k < j; ++k) dup = dup || cid[k] == fid[j];
              if (!dup) { cid[j] = fid[j]; nj = j + 1; }
            }
          }
          w = nj < ncand ? cw[nj] : (ncand == JB ? cw[JB - 1] + 1 : wend);     // (a repeated id stays where it is for the next chunk)
          if (nj == 0) break;                                     // (no context left)
          {
            int slot[JB];
#pragma unroll
            for (int j = 0; j < JB; ++j) slot[j] = hot.n_words ? hot.word_slot[j < nj ? cid[j] : id] : -1;
            // (the slots pinned in their registers HERE, once: behind the divergent "cached or not" branches below the compiler no longer
            // knows how many loads are in flight in front of a slot's and waits for everything, i.e. for the previous vector, at every test)
#pragma unroll
            for (int j = 0; j < JB; ++j) asm volatile("" : "+v"(slot[j]));
            // (cached vectors first, then the uncached ones: as "cached ? LDS : memory" per element both arms wrote one register, and
            // the lanes of the LDS arm waited for the other lanes' load from memory before every read)
#pragma unroll
            for (int j = 0; j < JB; ++j)
#pragma unroll
              for (int k = 0; k < CPL; ++k)
                if (j < nj && actk[k] && slot[j] >= 0) ctx[j][k] = locW[slot[j] * RS + l + k * GS];
#pragma unroll
            for (int j = 0; j < JB; ++j)
#pragma unroll
              for (int k = 0; k < CPL; ++k)
                if (j < nj && actk[k] && slot[j] < 0) ctx[j][k] = hog_load(a.param + (long long)cid[j] * dim + l + k * GS);
          }
          unsigned alive = (1u << nj) - 1u;
          // A visit's node update is issued at the head of the NEXT visit, behind that visit's wait for its node vector: vmcnt counts
          // loads, stores and atomics in one queue and, behind the divergent cached / uncached branches, the compiler waits for all of
          // them (vmcnt(0)) wherever it waits for a vector -- issued at the visit's end, an uncached node's device-scope atomic adds
          // were acknowledged (~2 k cycles) in front of the next visit's first multiply; issued here they have that visit's arithmetic
          // to be acknowledged in.  A path's nodes are distinct and the pending update is flushed before the chunk ends, so nothing
          // reads a row between its update's old and new place (one-stream Hogwild = the sequential pass: tests/test_gpu_w2v.py).
          int nd_pend = -1; double acc_pend[CPL];
#pragma unroll
          for (int k = 0; k < CPL; ++k) acc_pend[k] = 0.0;
          for (int c0 = hp0; c0 < hp1 && alive; c0 += GS) {
            const int n = hp1 - c0 < GS ? hp1 - c0 : GS;
            const int my_nd = c0 == hp0 ? h_nd0 : (l < n ? a.path_nodes[c0 + l] : 0);
            const int my_code = c0 == hp0 ? h_code0 : (l < n ? (int)a.path_codes[c0 + l] : 0);
            double pf[PF][CPL];                                   // node vectors requested PF nodes ahead
#pragma unroll
            for (int k = 0; k < PF; ++k) ld_node(__shfl(my_nd, gbase + (k < n ? k : 0), 64), k < n, pf[k]);
            for (int i = 0; i < n && alive; ++i) {
              const int nd = __shfl(my_nd, gbase + i, 64);
              const double one_minus_code = 1.0 - (double)__shfl(my_code, gbase + i, 64);
              double pvl[CPL], acc[CPL];
#pragma unroll
              for (int k = 0; k < CPL; ++k) { pvl[k] = pf[0][k]; acc[k] = 0.0; }
#pragma unroll
              for (int k = 0; k < CPL; ++k) asm volatile("" : "+v"(pvl[k]));       // (the vector is HERE: the wait stands in front of the update below)
              if (nd_pend >= 0) add_node(nd_pend, acc_pend);
#pragma unroll
              for (int q = 0; q + 1 < PF; ++q)
#pragma unroll
                for (int k = 0; k < CPL; ++k) pf[q][k] = pf[q + 1][k];
              {
                const int ia = i + PF;
                ld_node(__shfl(my_nd, gbase + (ia < n ? ia : 0), 64), ia < n, pf[PF - 1]);
              }
#pragma unroll
              for (int j = 0; j < JB; ++j) {
                if (alive & (1u << j)) {
                  double dot = ctx[j][0] * pvl[0];
#pragma unroll
                  for (int k = 1; k < CPL; ++k) dot += ctx[j][k] * pvl[k];
                  const double inner = group_sum64<GS>(dot);
                  if (inner <= -6.0 || inner >= 6.0) alive &= ~(1u << j);        // (quirk Q13: this pair's walk ends here)
                  else {
                    const double gg = (one_minus_code - sig_lookup(tab, inner)) * lr;
#pragma unroll
                    for (int k = 0; k < CPL; ++k) {
                      tmp[j][k] += gg * pvl[k];
                      pvl[k] += gg * ctx[j][k];                    // pv += g * ctx (optimizer.go:125): what pair j + 1 reads
                      acc[k] += gg * ctx[j][k];
                    }
                  }
                }
              }
              nd_pend = nd;
#pragma unroll
              for (int k = 0; k < CPL; ++k) acc_pend[k] = acc[k];
            }
          }
          if (nd_pend >= 0) add_node(nd_pend, acc_pend);
          {                                                        // ctx += tmp (model.go:74-76); the slots again in one round
            int slot[JB];
#pragma unroll
            for (int j = 0; j < JB; ++j) slot[j] = hot.n_words ? hot.word_slot[j < nj ? cid[j] : id] : -1;
#pragma unroll
            for (int j = 0; j < JB; ++j) asm volatile("" : "+v"(slot[j]));
#pragma unroll
            for (int j = 0; j < JB; ++j)
              if (j < nj) {
#pragma unroll
                for (int k = 0; k < CPL; ++k)
                  if (actk[k] && tmp[j][k] != 0.0) {
                    if (slot[j] >= 0) locW[slot[j] * RS + l + k * GS] += tmp[j][k];
                    else hog_add(a.param + (long long)cid[j] * dim + l + k * GS, tmp[j][k]);
                  }
              }
          }
        }
      }
      est += streams * a.est_scale;                                              // (observer estimate: see w2v_hogwild_kernel)
      if (est >= at + a.update_lr_batch) {
        do at += a.update_lr_batch; while (est >= at + a.update_lr_batch);
        if (lr < a.min_lr) lr = a.min_lr;
        else lr = a.init_lr * (1.0 - (double)at / (double)a.corpus_len);
      }
    }
    if ((pos + 1) % hot.merge_every == 0) merge();
  }
  merge();
  if (g == 0 && l == 0) *a.trained = a.n_words;
  if (g == streams - 1 && l == 0) *a.lr = lr;
}

__global__ void w2v_narrow_kernel(const double* p, long long n, float* out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (float)p[i];  // word2vec.go:315-318
}

// Huffman tree with the reference's tie-breaking (huffman.go:23-57): leaves stable-sorted by count; the
// merged node goes in front of every node of equal value.  Two queues: sorted leaves, and merged nodes
// (values non-decreasing) kept as runs of equal value that are consumed newest-first.
void build_huffman(const int64_t* counts, int64_t V, int max_depth, std::vector<long long>& off,
                   std::vector<int>& nodes, std::vector<unsigned char>& codes) {
  off.assign((size_t)V + 1, 0);
  nodes.clear(); codes.clear();
  if (V <= 0) return;
  const int64_t total = 2 * V - 1;
  std::vector<int64_t> val((size_t)total);
  std::vector<int> parent((size_t)total, -1);
  std::vector<unsigned char> code((size_t)total, 0);
  std::vector<int> order((size_t)V);
  std::iota(order.begin(), order.end(), 0);
  for (int64_t i = 0; i < V; ++i) val[i] = counts[i];
  {
    // stable sort of the leaves by count: LSD radix on the count alone (the indices start in order, every pass is stable) --
    // std::stable_sort with an indirect comparison took 70 of the 110 ms of the tree build at V = 10^6
    int64_t mx = 0;
    bool nonneg = true;
    for (int64_t i = 0; i < V; ++i) { mx = std::max(mx, counts[i]); nonneg = nonneg && counts[i] >= 0; }
    if (!nonneg || V < 4096) {
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return counts[x] < counts[y]; });
    } else {
      constexpr int RB = 11, RN = 1 << RB;
      std::vector<int> tmp((size_t)V);
      std::vector<int64_t> hist((size_t)RN);
      for (int shift = 0; shift < 63 && (mx >> shift) != 0; shift += RB) {
        std::fill(hist.begin(), hist.end(), 0);
        for (int64_t i = 0; i < V; ++i) hist[(size_t)((counts[order[i]] >> shift) & (RN - 1))]++;
        int64_t run = 0;
        for (int d = 0; d < RN; ++d) { const int64_t c = hist[d]; hist[d] = run; run += c; }
        for (int64_t i = 0; i < V; ++i) tmp[(size_t)hist[(size_t)((counts[order[i]] >> shift) & (RN - 1))]++] = order[i];
        order.swap(tmp);
      }
    }
  }
  // merged nodes, in creation order; a RUN = the merged nodes of one value (values are non-decreasing, so a run is a
  // contiguous range).  Only the last run grows, only the front run is consumed -- newest first (huffman.go inserts a merged
  // node in FRONT of every node of equal value).  Flat arrays: round 2 kept one std::vector per run, a heap allocation per
  // distinct merged value (10^6 of them on a Zipf tail).
  std::vector<int> mq((size_t)V);
  std::vector<int64_t> run_val; std::vector<int> run_beg, run_end;
  run_val.reserve(1 << 16); run_beg.reserve(1 << 16); run_end.reserve(1 << 16);
  size_t rfront = 0;
  int mq_n = 0;
  int64_t lq = 0;
  for (int64_t k = 0; k + 1 < V; ++k) {
    int pick[2];
    for (int s = 0; s < 2; ++s) {
      const bool have_leaf = lq < V, have_m = rfront < run_val.size();
      const bool take_m = have_leaf && have_m ? run_val[rfront] <= val[order[lq]] : have_m;
      if (take_m) {
        pick[s] = mq[--run_end[rfront]];
        if (rfront + 1 == run_val.size()) mq_n = run_end[rfront];          // (front run == last run: it is a plain stack)
        if (run_end[rfront] == run_beg[rfront]) ++rfront;
      } else {
        pick[s] = order[lq++];
      }
    }
    const int id = (int)(V + k);
    val[id] = val[pick[0]] + val[pick[1]];
    code[pick[0]] = 0; code[pick[1]] = 1;
    parent[pick[0]] = id; parent[pick[1]] = id;
    if (rfront < run_val.size() && run_val.back() == val[id]) { mq[mq_n++] = id; run_end.back() = mq_n; }
    else { run_val.push_back(val[id]); run_beg.push_back(mq_n); mq[mq_n++] = id; run_end.push_back(mq_n); }
  }
  // depth of every node: a parent is created after its children, so one pass from the root down
  std::vector<int> depth((size_t)total, 0);          // nodes on the leaf .. root chain, the node itself included
  for (int64_t i = total - 1; i >= 0; --i) depth[i] = parent[i] < 0 ? 1 : depth[parent[i]] + 1;
  // GetPath keeps cache[:depth] of the root-first chain (node.go:39-42): min(max_depth, len) - 1 (inner node, code) entries
  for (int64_t i = 0; i < V; ++i) {
    const int64_t d = std::min<int64_t>(max_depth, depth[i]);
    off[i + 1] = off[i] + (d > 0 ? d - 1 : 0);
  }
  nodes.resize((size_t)off[V]); codes.resize((size_t)off[V]);
  // fill: every leaf walks up to the root and writes its own range back to front -- independent per leaf, so in parallel
  // (leaves are visited in count order: neighbours in that order share most of their ancestors, so the parent[] walks
  // stay in cache; in id order every step of every walk was a miss)
  auto fill = [&](int64_t lo, int64_t hi) {
    for (int64_t ix = lo; ix < hi; ++ix) {
      const int64_t i = order[ix];
      const int len = depth[i];
      const int64_t keep = off[i + 1] - off[i];
      // chain (leaf .. root) position q = 0 .. len - 1; root-first index j = len - 1 - q; entry j (j < keep) = (chain[len-1-j] - V,
      // code[chain[len-2-j]]): walking up, at chain position q >= 1 we know chain[q] and its predecessor chain[q-1]
      int prev = (int)i;
      int p = parent[i];
      for (int q = 1; q < len; ++q) {
        const int j = len - 1 - q;
        if (j < keep) { nodes[(size_t)(off[i] + j)] = p - (int)V; codes[(size_t)(off[i] + j)] = code[prev]; }
        prev = p; p = parent[p];
      }
    }
  };
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt == 0 ? 1 : (nt > 16 ? 16 : nt);
  if (V < 20000 || nt == 1) fill(0, V);
  else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(fill, V * t / nt, V * (t + 1) / nt);
    for (auto& x : th) x.join();
  }
}

// vocabularies from 50 000 words on are built with the device (huffman.hip); below that the host builder is faster than the
// copies.  GOCTR_HUFFMAN_DEVICE=0 / 1 forces either.
bool huffman_on_device(int64_t V) {
  const char* f = getenv("GOCTR_HUFFMAN_DEVICE");
  if (f && *f) return *f != '0';
  return V >= 50000;
}

}  // namespace

struct goctr_w2v {
  goctr::Engine* const eng = &goctr::engine();   // the engine (device, streams, arena) the handle was created on
  goctr_w2v_cfg cfg{};
  int64_t V = 0;
  int64_t aux_rows = 0;
  DevBuf<double> param, aux, sigtab, lr, snap_param, snap_aux, touch_cnt;   // snap_*: the starting point of an exchange interval (multi-GPU); touch_cnt: ranks that updated a row
  DevBuf<double> hot_base;                                        // Hogwild kernel: the workgroups' base strips (HogHot::base)
  DevBuf<long long> path_off, trained, slice_idx, clip_lo, clip_hi;
  DevBuf<int> path_nodes, doc, hot_word_slot, hot_word_id;   // hot_*: the most frequent words, cached in LDS by the Hogwild kernel
  int n_hot_words = 0;
  std::vector<long long> h_counts;
  DevBuf<unsigned char> path_codes, keep;
  DevBuf<unsigned long long> lcg;
  std::vector<long long> h_off; std::vector<int> h_nodes; std::vector<unsigned char> h_codes;
  bool h_paths = false;          // the host copies above are filled (built on the host, or downloaded for goctr_w2v_get_paths)
  // single-call multi-device passes (cfg.devices = n): replicas on engines 1 .. n-1 (owned), `gen` counts what changed this
  // model's vectors from outside a multi-device pass, reps_gen = gen when the replicas were last known equal to it
  std::vector<goctr_w2v*> reps; uint64_t gen = 1, reps_gen = 0;
  long long path_total = 0;
  int64_t n_words = 0; bool has_keep = false;
  std::mutex mu;
};

namespace {

// Subsampler (modelutil/subsample/subsample.go:28-52): samples[id] = max(0, 1 - sqrt(threshold / cfs[id])) (raw counts,
// quirk Q14); a word is trained when samples[id] > u, u uniform in [0,1).  The reference draws u from Go's global
// math/rand stream, which cannot be regenerated outside Go; here u is a counter-based hash of (seed, position), so the
// mask is reproducible and the doc never leaves HBM.  Division and square root are correctly rounded on both sides, so
// samples[] itself is bit-identical.
__global__ void w2v_subsample_kernel(const int* doc, long long n, const long long* cfs, double threshold,
                                     unsigned long long seed, unsigned char* keep) {
  const long long pos = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pos >= n) return;
  double z = 1.0 - __dsqrt_rn(__ddiv_rn(threshold, (double)cfs[doc[pos]]));
  if (z < 0) z = 0;
  unsigned long long x = seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)(pos + 1);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
  x ^= x >> 27; x *= 0x94D049BB133111EBULL;
  x ^= x >> 31;
  const double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
  keep[pos] = z > u ? 1 : 0;
}

// data-parallel exchange (item2vec row): every rank trains its own corpus shard on a full replica; at an exchange
// the ranks' parameter DELTAS since the common snapshot are combined and applied to it, so all replicas agree again.
//   avg = false   p = p0 + sum_r d_r.  The deterministic single-stream mode (its tests compute the expected matrices from W
//                 single-device passes).  NOT usable for Hogwild training: the W ranks each walk the Huffman root (and every
//                 frequent node / word) thousands of times from the SAME stale snapshot, and stacking W such deltas is a step W
//                 times too long -- measured at cfg5, W = 8: HS loss 6.2 (once per pass) and 8.9 (every 10^5 words) against
//                 0.559 for the oracle's Hogwild run (round 5, profiles/r05_w2v_dp_exchange.txt; rounds 3-4 shipped this rule
//                 unmeasured).  Threads of the reference do not stack: each update reads the row the others have just written.
//   avg = true    p[row] = p0[row] + sum_r d_r[row] / #{r : d_r[row] != 0} -- per ROW the average over the ranks that updated it
//                 (local SGD / model averaging, which SURVEY 8(e) names, but a row only one rank trained keeps its whole
//                 update: rare words are not slowed down W times).  CPU simulation with the oracle's kernels (scripts/
//                 w2v_dp_sim.py, 10^7 words, W = 8): 0.5640 at 13 exchanges per pass, 0.6258 at one -- sequential pass 0.5590.
__global__ void w2v_delta_kernel(double* cur, const double* snap, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) cur[i] -= snap[i];
}
__global__ void w2v_apply_kernel(double* cur, const double* snap, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) cur[i] += snap[i];
}
// cnt[row] = 1 when this rank changed the row since the snapshot (cur already holds the delta)
__global__ void w2v_touched_kernel(const double* delta, long long rows, int dim, double* cnt) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  bool any = false;
  for (int c = 0; c < dim; ++c) any = any || delta[r * dim + c] != 0.0;
  cnt[r] = any ? 1.0 : 0.0;
}
__global__ void w2v_apply_avg_kernel(double* cur, const double* snap, const double* cnt, long long n, int dim) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double c = cnt[i / dim];
  cur[i] = snap[i] + (c > 1.0 ? cur[i] / c : cur[i]);
}

int exchange_deltas(goctr_w2v* w, bool avg) {
  Engine& e = engine();
  struct Part { DevBuf<double>* cur; DevBuf<double>* snap; long long rows; };
  Part parts[2] = {{&w->param, &w->snap_param, (long long)w->V}, {&w->aux, &w->snap_aux, (long long)w->aux_rows}};
  const int dim = w->cfg.dim;
  for (const Part& p : parts) {
    const long long n = p.rows * dim;
    if (n <= 0) continue;
    hipLaunchKernelGGL(w2v_delta_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, e.stream, p.cur->p, p.snap->p, n);
    GOCTR_HIP(hipGetLastError());
    if (avg) {
      if (w->touch_cnt.ensure((size_t)p.rows, false)) return -1;
      hipLaunchKernelGGL(w2v_touched_kernel, dim3((unsigned)cdiv(p.rows, 256)), dim3(256), 0, e.stream, p.cur->p, p.rows, dim, w->touch_cnt.p);
      GOCTR_HIP(hipGetLastError());
      if (comm_allreduce_f64_dev(w->touch_cnt.p, (size_t)p.rows)) return -1;
    }
    if (comm_allreduce_f64_dev(p.cur->p, (size_t)n)) return -1;
    if (avg) hipLaunchKernelGGL(w2v_apply_avg_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, e.stream, p.cur->p, p.snap->p, w->touch_cnt.p, n, dim);
    else hipLaunchKernelGGL(w2v_apply_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, e.stream, p.cur->p, p.snap->p, n);
    GOCTR_HIP(hipGetLastError());
  }
  return 0;
}

int env_int_w2v(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

int run_pass(goctr_w2v* w, int64_t corpus_len, double* lr_io) {
  Engine& e = engine();
  GOCTR_CHECK(w->n_words > 0, "goctr_w2v: no doc uploaded");
  if (w->lr.upload(lr_io, 1)) return -1;
  long long zero = 0;
  if (w->trained.upload(&zero, 1)) return -1;  // a fresh observer per iteration (word2vec.go:159-160)
  // data-parallel pass: the ranks exchange parameter deltas (p = p0 + sum_r (p_r - p0)) -- the Hogwild kernels every
  // `exchange_every` words per rank (default update_lr_batch = 10^5: SURVEY 8(e)), so that a rank sees the others' updates
  // during the pass like the reference's goroutines see each other's through the shared matrices (word2vec.go:198-243); the
  // deterministic single-stream kernel once per pass.  Every rank derives the same segment count from corpus_len.
  const bool dp = e.comm_active();
  int nseg = 1;
  if (dp && !w->cfg.deterministic) {
    // default interval: the observer's batch, but never so short that an exchange (a copy and an all-reduce of BOTH matrices and
    // their touched flags) moves more than ~1 KB per trained word -- V = 10^6 at D = 64 is 1 GB per exchange: every 10^5 words
    // would be all exchange; cfg5's 2.7 MB keep the 10^5
    const long long mat_bytes = ((long long)w->V + w->aux_rows) * w->cfg.dim * (long long)sizeof(double);
    const long long dflt = std::max<long long>(w->cfg.update_lr_batch, mat_bytes / 1024);
    const long long every = w->cfg.exchange_every == 0 ? dflt : w->cfg.exchange_every;
    if (every > 0) nseg = (int)std::min<long long>(4096, std::max<long long>(1, cdiv(cdiv(corpus_len, e.eff_world()), every)));
    nseg = std::max(1, env_int_w2v("GOCTR_W2V_SEGMENTS", nseg));
  }
  auto snapshot = [&]() -> int {
    const size_t np = (size_t)w->V * w->cfg.dim, na = (size_t)w->aux_rows * w->cfg.dim;
    if (w->snap_param.ensure(np, false) || (na && w->snap_aux.ensure(na, false))) return -1;
    GOCTR_HIP(hipMemcpyAsync(w->snap_param.p, w->param.p, np * sizeof(double), hipMemcpyDeviceToDevice, e.stream));
    if (na) GOCTR_HIP(hipMemcpyAsync(w->snap_aux.p, w->aux.p, na * sizeof(double), hipMemcpyDeviceToDevice, e.stream));
    return 0;
  };
  if (dp && w->cfg.deterministic && snapshot()) return -1;
  W2vDev a{};
  a.dim = w->cfg.dim; a.window = w->cfg.window; a.optimizer = w->cfg.optimizer; a.neg = w->cfg.neg_samples; a.model = w->cfg.model;
  a.init_lr = w->cfg.init_lr; a.min_lr = w->cfg.min_lr; a.update_lr_batch = w->cfg.update_lr_batch; a.V = w->V;
  a.param = w->param.p; a.aux = w->aux.p; a.path_off = w->path_off.p; a.path_nodes = w->path_nodes.p;
  a.path_codes = w->path_codes.p; a.sigtab = w->sigtab.p; a.doc = w->doc.p; a.keep = w->has_keep ? w->keep.p : nullptr;
  a.n_words = w->n_words; a.corpus_len = corpus_len; a.lr = w->lr.p; a.lcg = w->lcg.p; a.trained = w->trained.p;
  a.seg = 0; a.nseg = 1; a.est_scale = 1; a.seed_base = 0;
  if (w->cfg.deterministic) {
    hipLaunchKernelGGL(w2v_deterministic_kernel, dim3(1), dim3(64), 0, e.stream, a);
    GOCTR_HIP(hipGetLastError());
  } else {
    int streams = w->cfg.streams > 0 ? w->cfg.streams : 8192;
    if ((int64_t)streams > w->n_words) streams = (int)w->n_words;
    // Every stream is a Hogwild worker and the hot rows of the workgroups' LDS copies are AVERAGED at the merges: a worker
    // that walks only a few dozen positions contributes a few dozen updates' worth of learning to them however many workers
    // there are (V = 10^6, D = 64, 10^6 words on 32 768 streams: HS loss 0.665 against the oracle's 0.624).  So the
    // parallelism is capped by the corpus: at least 256 positions per stream (bench.py's 10^7 words / 32 768 streams = 305).
    // Data-parallel passes take 2048: every stream of every rank starts an exchange interval from the same stale snapshot, and
    // what the ranks learn beside each other is AVERAGED at the exchange (exchange_deltas), so the learning of an interval is
    // what ONE rank's streams make of it -- more, shorter streams per rank make less of it.  Measured at cfg5 on W = 8 ranks
    // (profiles/r05_w2v_dp_gpu_sweep.txt; oracle 0.5587): 256 positions per stream 0.5962, 1024 0.5738, 4096 0.5679.
    const int min_pos = dp ? 2048 : 256;
    {
      const int64_t cap = std::max<int64_t>(1, w->n_words / min_pos);
      if ((int64_t)streams > cap) streams = (int)cap;
    }
    // slices = the reference's goroutines (window-clipping units); 0: one slice per stream (every piece clips its own windows)
    int slices = w->cfg.slices > 0 ? std::min(w->cfg.slices, streams) : streams;
    const int per = streams / slices;                 // workers per slice (the last slice takes the remainder)
    // IndexPerThread (modelutil.go:32-41) over the slices, then each slice cut evenly among its workers
    std::vector<long long> sidx((size_t)slices + 1);
    sidx[0] = 0; sidx[slices] = w->n_words;
    for (int i = 1; i < slices; ++i) sidx[i] = sidx[i - 1] + (long long)std::trunc((double)((w->n_words + i) / slices));
    std::vector<long long> idx((size_t)streams + 1), clo((size_t)streams), chi((size_t)streams);
    int g = 0;
    for (int sl = 0; sl < slices; ++sl) {
      const int nw = sl + 1 < slices ? per : streams - g;
      const long long a0 = sidx[sl], a1 = sidx[sl + 1];
      for (int k = 0; k < nw; ++k, ++g) {
        idx[g] = a0 + (a1 - a0) * k / nw;
        clo[g] = a0; chi[g] = a1;
      }
    }
    idx[streams] = w->n_words;
    if (w->slice_idx.alloc(idx.size(), false) || w->slice_idx.upload(idx.data(), idx.size())) return -1;
    if (w->clip_lo.alloc(clo.size(), false) || w->clip_lo.upload(clo.data(), clo.size())) return -1;
    if (w->clip_hi.alloc(chi.size(), false) || w->clip_hi.upload(chi.data(), chi.size())) return -1;
    const int dim = w->cfg.dim;
    const int dimr = dim <= 8 ? 8 : dim <= 16 ? 16 : dim <= 32 ? 32 : 64;     // components per lane group, padded
    // kernel variant.  Skip-gram + HS runs the node-major kernel (w2v_hogwild_nm_kernel): JB pairs per chunk, CPL components per
    // lane, WPS wavefronts per SIMD, PF nodes requested ahead; GOCTR_W2V_JB=0 (and CBOW / negative sampling) the pair-major one
    const bool sg_hs = w->cfg.model == 0 && w->cfg.optimizer == 0;
    // Measured at V = 10 681, dim 16, 10^7 words, 32 768 streams (scripts/w2v_jb.sh, profiles/r04_w2v_node_major.txt):
    //   pair-major, ds_bpermute sums (round 3 .. 4 start)   235 M words/s   5.04 KB/word memory-side   HS loss 0.5684
    //   pair-major, DPP sums                                263 M           4.68
    //   JB 3, 16 lanes x 1, WPS 8, PF 2                     375 M           3.08                       0.5684
    //   JB 4,  8 lanes x 2, WPS 4, PF 8, 384 hot rows       426 M           1.55                       0.5695
    //   JB 4,  8 lanes x 2, WPS 4, PF 8, 256 hot rows       412 M                                      0.5654   <- default
    const int jb = sg_hs ? env_int_w2v("GOCTR_W2V_JB", dimr >= 16 ? 4 : 3) : 0;
    const int cpl = jb > 0 && dimr >= 16 ? 2 : 1;
    const int wps = jb > 0 && cpl > 1 ? 4 : 8;
    const int pf = cpl > 1 ? 8 : 2;
    const int GSr = dimr / (cpl > 1 ? 2 : 1);                                  // lanes per group
    const int HOT = jb > 0 ? hog_hot_doubles(wps) : HOG_HOT_DOUBLES;
    // hot rows cached in LDS per workgroup: the heaviest Huffman nodes are the LAST merges (weights are non-decreasing
    // along the merge order), the hottest words the most frequent ones (ties: lower id first)
    // GOCTR_W2V_HOT: 0 = no hot rows; 1 (default) = three quarters of what the LDS tables hold (192 rows at dim 16; WPS 4, whose
    // tables are twice as large: half, 256 rows -- the table above); n > 1 = at most n rows per table.  Measured at V = 10 681, 10^7 words (scripts/w2v_hot.sh, profiles/r04_w2v_hot_set.txt): 128 rows 189 M
    // words/s, 5.8 KB/word memory-side, HS loss 0.5651 (oracle 0.557-0.560); 192 rows 231 M, loss 0.5686; 256 rows 232 M, 4.8 KB,
    // loss 0.5716 -- hot rows are AVERAGED over the workgroups (see above), so every row that joins the hot set learns more slowly:
    // the last quarter buys 0.6 % of speed for 0.5 % of loss, against a 3 % gate.
    const int hot_knob = env_int_w2v("GOCTR_W2V_HOT", 1);
    const int hot_cap = HOT / dimr;
    const int rows_cached = hot_knob == 0 ? 0 : (hot_knob == 1 ? (jb > 0 && wps <= 4 ? hot_cap / 2 : hot_cap * 3 / 4) : std::min(hot_knob, hot_cap));
    HogHot hot{};
    hot.n_nodes = w->cfg.optimizer == 0 ? (int)std::min<int64_t>(rows_cached, w->aux_rows) : 0;
    hot.node0 = w->aux_rows - hot.n_nodes;
    if (w->n_hot_words != std::min<int64_t>(rows_cached, w->V) || !w->hot_word_slot.p) {
      const int nh = (int)std::min<int64_t>(rows_cached, w->V);
      std::vector<int> order((size_t)w->V), slot((size_t)w->V, -1), ids((size_t)std::max(nh, 1), 0);
      std::iota(order.begin(), order.end(), 0);
      if (!w->h_counts.empty())
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return w->h_counts[x] > w->h_counts[y]; });
      for (int k = 0; k < nh; ++k) { slot[order[k]] = k; ids[k] = order[k]; }
      if (w->hot_word_slot.alloc(slot.size(), false) || w->hot_word_slot.upload(slot.data(), slot.size())) return -1;
      if (w->hot_word_id.alloc(ids.size(), false) || w->hot_word_id.upload(ids.data(), ids.size())) return -1;
      w->n_hot_words = nh;
    }
    hot.n_words = w->n_hot_words; hot.word_slot = w->hot_word_slot.p; hot.word_id = w->hot_word_id.p;
    hot.merge_every = 32;   // (round 4: 16 -> 32 with the larger hot set: 229 -> 235 M words/s, same loss; 64: no further gain)
    const int nwg = (int)cdiv(streams, HOG_THREADS / GSr);
    if (w->hot_base.ensure((size_t)nwg * 2 * HOT, false)) return -1;
    hot.base = w->hot_base.p;
    hot.merge_scale = 1.0 / (double)nwg;
    {
      long long longest = 0;
      for (int k = 0; k < streams; ++k) longest = std::max(longest, idx[k + 1] - idx[k]);
      GOCTR_CHECK(longest < (1LL << 31), "goctr_w2v: a stream's piece of %lld words (more streams, or a shorter doc)", longest);
    }
    if (dp) {
      // the observer's estimate counts the other ranks' words too (a rank that counted only its own shard let the rate
      // decay world times too slowly), and the ranks' streams draw their window shrinks from different seeds
      // (seed ranges: a FIXED stride per rank -- shards of unequal length give the ranks different stream counts, and rank x
      // streams would overlap, ADVICE r5.  The estimate stays this rank's words x world: the shards end at the reference's slice
      // boundaries, a few words apart)
      a.est_scale = e.eff_world(); a.seed_base = (long long)e.rank << 20;
    }
    a.nseg = nseg;
    for (int seg = 0; seg < nseg; ++seg) {
    a.seg = seg;
    hot.max_len = 0;          // the longest part any stream walks in this launch (all threads of a workgroup loop alike: merges)
    for (int k = 0; k < streams; ++k) {
      const long long len0 = idx[k + 1] - idx[k];
      hot.max_len = std::max(hot.max_len, len0 * (seg + 1) / nseg - len0 * seg / nseg);
    }
    if (dp && snapshot()) return -1;
    const dim3 grid((unsigned)nwg), block(HOG_THREADS);
#define GOCTR_HOG_ARGS 0, e.stream, a, streams, w->slice_idx.p, w->clip_lo.p, w->clip_hi.p, hot
    bool launched = false;
#define GOCTR_NM(GS, CPL, JB, WPS, PF)                                                                              \
    if (!launched && GSr == GS && cpl == CPL && jb == JB && wps == WPS && pf == PF) {                                \
      hipLaunchKernelGGL((w2v_hogwild_nm_kernel<GS, CPL, JB, WPS, PF>), grid, block, GOCTR_HOG_ARGS); launched = true; \
    }
    if (jb > 0) {
      GOCTR_NM(8, 2, 4, 4, 8) GOCTR_NM(16, 2, 4, 4, 8) GOCTR_NM(32, 2, 4, 4, 8)      // dim <= 16 / 32 / 64: the default shapes
      // (round 6, VERDICT r5 item 5 -- the 2-pair chunk on 8 wavefronts per SIMD, <8, 2, 2, 8, 4>: the compiler's resource report at
      // 64 registers is 63 spilled (the 1-pair chunk: 41; round 4's 4-pair chunk: 38-43) -- the walk's addresses, stream state and
      // float64 pairs alone exceed 64 registers, so 8 wavefronts per SIMD would run out of scratch memory; profiles/r06_w2v_regs.txt)
      GOCTR_NM(8, 1, 3, 8, 2)                                                      // dim <= 8
      GOCTR_NM(16, 1, 3, 8, 2) GOCTR_NM(8, 2, 4, 4, 4)                             // (A/B: one component per lane; shallower prefetch)
      GOCTR_CHECK(launched, "goctr_w2v: no node-major kernel for lanes %d x %d components, JB %d, WPS %d, PF %d", GSr, cpl, jb, wps, pf);
    }
#undef GOCTR_NM
#define GOCTR_HOG_MO(GS, M, O) hipLaunchKernelGGL((w2v_hogwild_kernel<GS, M, O>), grid, block, GOCTR_HOG_ARGS)
#define GOCTR_HOG(GS)                                                   \
  do {                                                                  \
    if (a.model == 1) { if (a.optimizer == 0) GOCTR_HOG_MO(GS, 1, 0); else GOCTR_HOG_MO(GS, 1, 1); } \
    else { if (a.optimizer == 0) GOCTR_HOG_MO(GS, 0, 0); else GOCTR_HOG_MO(GS, 0, 1); }              \
  } while (0)
    if (launched) {}
    else if (dim <= 8) GOCTR_HOG(8);
    else if (dim <= 16) GOCTR_HOG(16);
    else if (dim <= 32) GOCTR_HOG(32);
    else GOCTR_HOG(64);
#undef GOCTR_HOG
#undef GOCTR_HOG_MO
#undef GOCTR_HOG_ARGS
    GOCTR_HIP(hipGetLastError());
    if (dp && exchange_deltas(w, true)) return -1;
    }   // segments
  }
  if (dp && w->cfg.deterministic && exchange_deltas(w, false)) return -1;
  GOCTR_HIP(hipStreamSynchronize(e.stream));
  return w->lr.download(lr_io, 1);
}

}  // namespace

// ---- single-call multi-device passes (goctr_w2v_cfg::devices)
// cut[r] .. cut[r + 1] = rank r's words: slices [r, r + 1) * S / N of IndexPerThread's cut (modelutil.go:32-41) when the slices
// divide evenly, else an equal contiguous range
static void w2v_shard_cuts(long long n_words, int S, int N, long long* cut) {
  cut[0] = 0; cut[N] = n_words;
  if (S >= N && S % N == 0) {
    std::vector<long long> sidx((size_t)S + 1, 0);
    sidx[S] = n_words;
    for (int i = 1; i < S; ++i) sidx[i] = sidx[i - 1] + (long long)std::trunc((double)((n_words + i) / S));
    for (int r = 1; r < N; ++r) cut[r] = sidx[(size_t)r * (S / N)];
  } else {
    for (int r = 1; r < N; ++r) cut[r] = n_words * r / N;
  }
}
static int w2v_upload_one(goctr_w2v* w, const int32_t* doc, int64_t n_words, const uint8_t* keep_mask) {
  if (w->doc.alloc((size_t)n_words, false) || w->doc.upload(doc, (size_t)n_words)) return -1;
  w->has_keep = keep_mask != nullptr;
  if (keep_mask && (w->keep.alloc((size_t)n_words, false) || w->keep.upload(keep_mask, (size_t)n_words))) return -1;
  w->n_words = n_words;
  return 0;
}
// replicas on engines 1 .. n-1 (same cfg, same counts: the same tree) and the doc cut at the reference's slice boundaries;
// fill(rank's handle, rank, first word, end word) puts the rank's shard in place (called with the rank's engine bound and locked)
template <class Fill>
static int w2v_multi_shards(goctr_w2v* w, int64_t n_words, Fill fill) {
  const int N = w->cfg.devices;
  Engine* e0 = engine_at(0);
  GOCTR_CHECK(N == engine_count() && e0 && e0->world == N && (e0->loop || e0->nccl_comm),
              "w2v cfg.devices = %d, but goctr_init_devices set up %d engine(s)", N, (e0 && (e0->loop || e0->nccl_comm)) ? e0->world : 1);
  GOCTR_CHECK(w->eng == e0, "multi-device item2vec: the handle must live on engine 0");
  GOCTR_CHECK(n_words >= N, "multi-device item2vec: %lld words for %d ranks", (long long)n_words, N);
  if ((int)w->reps.size() != N) { for (auto* r : w->reps) goctr_w2v_destroy(r); w->reps.assign((size_t)N, nullptr); w->reps_gen = 0; }
  // rank r takes slices [r, r + 1) * S / N of IndexPerThread's cut (modelutil.go:32-41) when the slices divide evenly, else an
  // equal contiguous range; inside its shard every rank cuts its own S / N slices
  const int S = w->cfg.slices;
  std::vector<long long> cut((size_t)N + 1, 0);
  w2v_shard_cuts(n_words, S, N, cut.data());
  goctr_w2v_cfg rc = w->cfg;
  rc.devices = 0; rc.slices = S > 0 ? std::max(1, S / N) : 0;
  for (int k = N - 1; k >= 0; --k) {           // (rank 0 last: a device-resident source keeps its prefix)
    Engine* ek = engine_at(k);
    EngineScope on(ek);
    std::lock_guard<std::recursive_mutex> elk(ek->mu);
    if (k > 0 && !w->reps[k]) {
      std::vector<int64_t> counts(w->h_counts.begin(), w->h_counts.end());
      if (goctr_w2v_create(&rc, w->V, counts.data(), &w->reps[k])) return -1;
      w->reps_gen = 0;
    }
    if (fill(k == 0 ? w : w->reps[k], k, cut[k], cut[k + 1])) return -1;
  }
  return 0;
}
static int w2v_multi_pass(goctr_w2v* w, int64_t corpus_len, double* lr) {
  const int N = w->cfg.devices;
  GOCTR_CHECK((int)w->reps.size() == N, "multi-device item2vec: upload the doc first (goctr_w2v_upload_doc / goctr_w2v_train)");
  if (comm_group_reset()) return -1;
  const bool sync = w->reps_gen != w->gen;
  const double lr_in = *lr;
  const int slices0 = w->cfg.slices;
  w->cfg.slices = slices0 > 0 ? std::max(1, slices0 / N) : 0;       // (rank 0 cuts its shard like the replicas do)
  std::vector<double> lrs((size_t)N, lr_in);
  const int rc = run_on_engines(N, [&](int k) -> int {
    Engine& e = engine();
    std::lock_guard<std::recursive_mutex> elk(e.mu);
    const bool prev = e.comm_enabled;
    e.comm_enabled = true;
    goctr_w2v* wk = k == 0 ? w : w->reps[k];
    std::unique_lock<std::mutex> lk(wk->mu, std::defer_lock);
    if (k > 0) lk.lock();
    int r = 0;
    if (sync) r = comm_broadcast(wk->param.p, sizeof(double) * (size_t)w->V * w->cfg.dim, 0) ||
                  comm_broadcast(wk->aux.p, sizeof(double) * (size_t)w->aux_rows * w->cfg.dim, 0);
    if (!r) r = run_pass(wk, corpus_len, &lrs[(size_t)k]);
    if (r) { const std::string msg = goctr_last_error(); comm_abort_on_failure(); set_error("%s", msg.c_str()); }
    e.comm_enabled = prev;
    return r;
  });
  w->cfg.slices = slices0;
  if (rc) { w->reps_gen = 0; return -1; }
  w->reps_gen = w->gen;
  *lr = lrs[0];
  return 0;
}

extern "C" {

void goctr_w2v_cfg_default(goctr_w2v_cfg* c) {
  memset(c, 0, sizeof *c);  // options.go:38-58 + wordemb.go:10-18
  c->dim = 16; c->window = 5; c->optimizer = 0; c->model = 0; c->neg_samples = 5;
  c->init_lr = 0.025; c->min_lr = 0.025 * 1.0e-4; c->update_lr_batch = 100000; c->max_depth = 100;
  c->deterministic = 0; c->streams = 8192; c->slices = 16;
}

int goctr_w2v_create(const goctr_w2v_cfg* cfg, int64_t V, const int64_t* counts, goctr_w2v** out) {
  GOCTR_ENTER();
  GOCTR_CHECK(cfg && counts && out && V > 0, "goctr_w2v_create: bad arguments");
  GOCTR_CHECK(cfg->dim > 0 && cfg->dim <= 64, "goctr_w2v: dim %d not in 1..64", cfg->dim);
  GOCTR_CHECK(cfg->window > 0 && cfg->max_depth > 0 && cfg->update_lr_batch > 0, "goctr_w2v: bad options");
  GOCTR_CHECK(cfg->model == 0 || cfg->model == 1, "goctr_w2v: model must be skip-gram (0) or cbow (1)");
  GOCTR_CHECK(cfg->optimizer == 0 || cfg->optimizer == 1, "goctr_w2v: optimizer must be hs (0) or ns (1)");
  std::unique_ptr<goctr_w2v> w(new goctr_w2v);
  w->cfg = *cfg; w->V = V;
  w->h_counts.assign(counts, counts + V);
  w->aux_rows = cfg->optimizer == 0 ? std::max<int64_t>(V - 1, 1) : V;
  if (w->param.alloc((size_t)V * cfg->dim) || w->aux.alloc((size_t)w->aux_rows * cfg->dim)) return -1;
  if (huffman_on_device(V)) {
    // large vocabularies: sort and path fill on the device, the merge on the host in sorted-rank space (huffman.hip); the
    // paths are born in HBM and reach the host only if goctr_w2v_get_paths asks for them
    if (huffman_build_device(w->h_counts.data(), V, cfg->max_depth, w->path_off, w->path_nodes, w->path_codes, &w->path_total, nullptr)) return -1;
  } else {
    build_huffman(counts, V, cfg->max_depth, w->h_off, w->h_nodes, w->h_codes);
    GOCTR_CHECK(w->h_nodes.size() < ((size_t)1 << 31), "goctr_w2v: Huffman paths with 2^31 entries or more (the Hogwild walk indexes them with 32 bits)");
    w->h_paths = true; w->path_total = (long long)w->h_nodes.size();
    if (w->path_off.alloc(w->h_off.size(), false) || w->path_off.upload(w->h_off.data(), w->h_off.size())) return -1;
    if (w->path_nodes.alloc(std::max<size_t>(w->h_nodes.size(), 1)) ||
        (!w->h_nodes.empty() && w->path_nodes.upload(w->h_nodes.data(), w->h_nodes.size()))) return -1;
    if (w->path_codes.alloc(std::max<size_t>(w->h_codes.size(), 1)) ||
        (!w->h_codes.empty() && w->path_codes.upload(w->h_codes.data(), w->h_codes.size()))) return -1;
  }
  std::vector<double> tab(1000);
  for (int i = 0; i < 1000; ++i) {  // sigmoid_table.go:28-38
    const double ev = std::exp(((double)i / 1000.0 * 2. - 1.) * 6.0);
    tab[i] = ev / (ev + 1.);
  }
  if (w->sigtab.alloc(1000, false) || w->sigtab.upload(tab.data(), 1000)) return -1;
  unsigned long long one = 1;  // modelutil.go:21-23: next starts at 1
  if (w->lcg.alloc(1, false) || w->lcg.upload(&one, 1)) return -1;
  if (w->lr.alloc(1) || w->trained.alloc(1)) return -1;
  *out = w.release();
  return 0;
}

void goctr_w2v_destroy(goctr_w2v* w) {
  if (!w) return;
  for (goctr_w2v* r : w->reps) goctr_w2v_destroy(r);
  delete w;
}

int goctr_w2v_set_param(goctr_w2v* w, const double* param) {
  GOCTR_ENTER_H(w);
  GOCTR_CHECK(w && param, "goctr_w2v_set_param: null argument");
  ++w->gen;
  return w->param.upload(param, (size_t)w->V * w->cfg.dim);
}
int goctr_w2v_set_aux(goctr_w2v* w, const double* aux) {
  GOCTR_ENTER_H(w);
  GOCTR_CHECK(w && aux, "goctr_w2v_set_aux: null argument");
  ++w->gen;
  return w->aux.upload(aux, (size_t)w->aux_rows * w->cfg.dim);
}
int goctr_w2v_get_param(goctr_w2v* w, double* param) {
  GOCTR_ENTER_H(w);
  GOCTR_CHECK(w && param, "goctr_w2v_get_param: null argument");
  return w->param.download(param, (size_t)w->V * w->cfg.dim);
}
int goctr_w2v_get_aux(goctr_w2v* w, double* aux) {
  GOCTR_ENTER_H(w);
  GOCTR_CHECK(w && aux, "goctr_w2v_get_aux: null argument");
  return w->aux.download(aux, (size_t)w->aux_rows * w->cfg.dim);
}

int goctr_w2v_get_paths(goctr_w2v* w, int64_t* path_off, int32_t* nodes, uint8_t* codes, int64_t cap, int64_t* total) {
  GOCTR_ENTER_H(w);
  GOCTR_CHECK(w, "goctr_w2v_get_paths: null argument");
  std::lock_guard<std::mutex> lk(w->mu);
  if (!w->h_paths) {       // built on the device: fetched on first request
    w->h_off.resize((size_t)w->V + 1); w->h_nodes.resize((size_t)w->path_total); w->h_codes.resize((size_t)w->path_total);
    if (w->path_off.download(w->h_off.data(), w->h_off.size())) return -1;
    if (w->path_total && (w->path_nodes.download(w->h_nodes.data(), w->h_nodes.size()) || w->path_codes.download(w->h_codes.data(), w->h_codes.size()))) return -1;
    w->h_paths = true;
  }
  if (total) *total = (int64_t)w->h_nodes.size();
  if (path_off) for (size_t i = 0; i < w->h_off.size(); ++i) path_off[i] = w->h_off[i];
  const int64_t n = std::min<int64_t>(cap, (int64_t)w->h_nodes.size());
  if (nodes) memcpy(nodes, w->h_nodes.data(), sizeof(int32_t) * (size_t)n);
  if (codes) memcpy(codes, w->h_codes.data(), (size_t)n);
  return 0;
}

int goctr_huffman_build(const int64_t* counts, int64_t V, int max_depth, int64_t* path_off, int32_t* nodes, uint8_t* codes,
                        int64_t cap, int64_t* total, double* build_ms) {
  GOCTR_CHECK(counts && V > 0 && path_off && max_depth > 0, "goctr_huffman_build: bad arguments");
  GOCTR_CHECK(V <= 0x3fffffff, "goctr_huffman_build: V = %lld exceeds the 2^30 words the int32 node ids can number", (long long)V);
  if (engine().inited && huffman_on_device(V)) {
    // with a device bound: the build of huffman.hip.  *build_ms = until the paths are resident in HBM (what goctr_w2v_create
    // pays); copying them out to the caller's arrays (1.2 GB at V = 10^7) comes on top and is not part of the build
    GOCTR_ENTER();
    DevBuf<long long> off; DevBuf<int> nd; DevBuf<unsigned char> cd;
    long long tot = 0; double parts[4] = {0, 0, 0, 0};
    std::vector<long long> c64(counts, counts + V);
    if (huffman_build_device(c64.data(), V, max_depth, off, nd, cd, &tot, parts)) return -1;
    if (build_ms) *build_ms = parts[3];
    std::vector<long long> ho((size_t)V + 1);
    if (off.download(ho.data(), ho.size())) return -1;
    for (size_t i = 0; i < ho.size(); ++i) path_off[i] = ho[i];
    if (total) *total = tot;
    const int64_t n = std::min<int64_t>(cap, tot);
    if (nodes && n > 0 && nd.download(nodes, (size_t)n)) return -1;
    if (codes && n > 0 && cd.download(codes, (size_t)n)) return -1;
    return 0;
  }
  std::vector<long long> off;
  std::vector<int> nd;
  std::vector<unsigned char> cd;
  const auto t0 = std::chrono::steady_clock::now();
  build_huffman(counts, V, max_depth, off, nd, cd);
  if (build_ms) *build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (size_t i = 0; i < off.size(); ++i) path_off[i] = off[i];
  if (total) *total = (int64_t)nd.size();
  const int64_t n = std::min<int64_t>(cap, (int64_t)nd.size());
  if (nodes && n > 0) memcpy(nodes, nd.data(), sizeof(int32_t) * (size_t)n);
  if (codes && n > 0) memcpy(codes, cd.data(), (size_t)n);
  return 0;
}

int goctr_w2v_shard_cuts(int64_t n_words, int slices, int devices, int64_t* cuts) {
  GOCTR_CHECK(cuts && devices >= 1 && n_words >= devices && slices >= 0, "goctr_w2v_shard_cuts: bad arguments");
  std::vector<long long> c((size_t)devices + 1);
  w2v_shard_cuts(n_words, slices, devices, c.data());
  for (int r = 0; r <= devices; ++r) cuts[r] = c[(size_t)r];
  return 0;
}

int goctr_w2v_upload_doc(goctr_w2v* w, const int32_t* doc, int64_t n_words, const uint8_t* keep_mask) {
  GOCTR_ENTER_H(w);
  GOCTR_CHECK(w && doc && n_words > 0, "goctr_w2v_upload_doc: bad arguments");
  std::lock_guard<std::mutex> lk(w->mu);
  for (int64_t i = 0; i < n_words; ++i)
    GOCTR_CHECK(doc[i] >= 0 && doc[i] < w->V, "doc[%lld] = %d outside the dictionary (V = %lld)", (long long)i, doc[i], (long long)w->V);
  if (w->cfg.devices > 1)
    return w2v_multi_shards(w, n_words, [&](goctr_w2v* wk, int, long long lo, long long hi) {
      return w2v_upload_one(wk, doc + lo, hi - lo, keep_mask ? keep_mask + lo : nullptr);
    });
  return w2v_upload_one(w, doc, n_words, keep_mask);
}

int goctr_w2v_train_resident(goctr_w2v* w, int64_t corpus_len, double* lr) {
  GOCTR_ENTER_H(w);
  GOCTR_CHECK(w && lr && corpus_len > 0, "goctr_w2v_train_resident: bad arguments");
  std::lock_guard<std::mutex> lk(w->mu);
  if (w->cfg.devices > 1) return w2v_multi_pass(w, corpus_len, lr);
  return run_pass(w, corpus_len, lr);
}

int goctr_w2v_train(goctr_w2v* w, const int32_t* doc, int64_t n_words, int64_t corpus_len, const uint8_t* keep_mask,
                    double* lr) {
  if (goctr_w2v_upload_doc(w, doc, n_words, keep_mask)) return -1;
  return goctr_w2v_train_resident(w, corpus_len, lr);
}

// word2vec.Train's prelude over a device-resident corpus (word2vec.go:90-135): the model is sized by the corpus'
// dictionary, the Huffman tree / NS table come from its cfs.
int goctr_w2v_create_from_corpus(const goctr_w2v_cfg* cfg, goctr_corpus* c, goctr_w2v** out) {
  GOCTR_ENTER_H(c);
  GOCTR_CHECK(cfg && c && out, "goctr_w2v_create_from_corpus: null argument");
  std::vector<int64_t> cfs;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    GOCTR_CHECK(c->built, "goctr_w2v_create_from_corpus: call goctr_corpus_build first");
    cfs.resize((size_t)c->V);
    if (c->cfs.download(reinterpret_cast<long long*>(cfs.data()), cfs.size())) return -1;
  }
  return goctr_w2v_create(cfg, (int64_t)cfs.size(), cfs.data(), out);
}

// The training doc of one iteration = the corpus' IndexedDoc (device-to-device) + a fresh subsampling mask.
int goctr_w2v_use_corpus(goctr_w2v* w, goctr_corpus* c, double subsample_threshold, uint64_t seed) {
  GOCTR_ENTER_H(w);
  GOCTR_CHECK(w && c, "goctr_w2v_use_corpus: null argument");
  std::lock_guard<std::mutex> lk(w->mu);
  std::lock_guard<std::mutex> lk2(c->mu);
  GOCTR_CHECK(c->built && c->V == w->V, "goctr_w2v_use_corpus: corpus not built or dictionary size %lld != model V %lld",
              (long long)c->V, (long long)w->V);
  GOCTR_CHECK(c->n_indexed > 0, "goctr_w2v_use_corpus: every word was filtered out");
  const long long n = c->n_indexed;
  hipStream_t s = engine().stream;
  if (w->doc.ensure((size_t)n, false)) return -1;
  GOCTR_HIP(hipMemcpyAsync(w->doc.p, c->indexed.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s));
  w->has_keep = subsample_threshold >= 0;
  if (w->has_keep) {
    if (w->keep.ensure((size_t)n, false)) return -1;
    hipLaunchKernelGGL(w2v_subsample_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, w->doc.p, n, c->cfs.p,
                       subsample_threshold, (unsigned long long)seed, w->keep.p);
    GOCTR_HIP(hipGetLastError());
  }
  w->n_words = n;
  if (w->cfg.devices > 1) {
    // cfg.devices = n: the doc and its mask were made on engine 0; ranks 1 .. n-1 take their shards device to device, rank 0
    // keeps the prefix of what it holds
    GOCTR_HIP(hipStreamSynchronize(s));
    const goctr_w2v* src = w;
    const int dev0 = w->eng->device;
    return w2v_multi_shards(w, n, [&](goctr_w2v* wk, int k, long long lo, long long hi) -> int {
      wk->n_words = hi - lo;
      wk->has_keep = src->has_keep;
      if (k == 0) return 0;
      Engine& ek = engine();
      if (wk->doc.ensure((size_t)(hi - lo), false) || (src->has_keep && wk->keep.ensure((size_t)(hi - lo), false))) return -1;
      GOCTR_HIP(hipMemcpyPeerAsync(wk->doc.p, ek.device, src->doc.p + lo, dev0, sizeof(int) * (size_t)(hi - lo), ek.stream));
      if (src->has_keep) GOCTR_HIP(hipMemcpyPeerAsync(wk->keep.p, ek.device, src->keep.p + lo, dev0, (size_t)(hi - lo), ek.stream));
      GOCTR_HIP(hipStreamSynchronize(ek.stream));
      return 0;
    });
  }
  return 0;
}

int goctr_w2v_get_keep_mask(goctr_w2v* w, uint8_t* keep, int64_t n) {
  GOCTR_ENTER_H(w);
  GOCTR_CHECK(w && keep, "goctr_w2v_get_keep_mask: null argument");
  std::lock_guard<std::mutex> lk(w->mu);
  GOCTR_CHECK(w->has_keep && n == w->n_words, "goctr_w2v_get_keep_mask: no mask resident or %lld != %lld words", (long long)n, (long long)w->n_words);
  return w->keep.download(keep, (size_t)n);
}

int goctr_w2v_export_f32(goctr_w2v* w, float* out) {
  GOCTR_ENTER_H(w);
  GOCTR_CHECK(w && out, "goctr_w2v_export_f32: null argument");
  const long long n = (long long)w->V * w->cfg.dim;
  DevBuf<float> d;
  if (d.alloc((size_t)n, false)) return -1;
  hipLaunchKernelGGL(w2v_narrow_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, engine().stream, w->param.p, n, d.p);
  GOCTR_HIP(hipGetLastError());
  return d.download(out, (size_t)n);
}

}  // extern "C"
