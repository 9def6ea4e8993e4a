// search.hip -- embedding k-NN search: brute-force cosine top-k over all items, float64.
//
//   search.go:92-134   Searcher.Search     -> goctr_searcher_search (Q queries per call)
//   searchutil.go:17-26 Cosine              -> score = dot / n1 / n2, dot accumulated in index order (bit-exact)
//   embutil.go:21-27   Norm                -> knn_norm_kernel (sequential sum, IEEE sqrt)
//
// The reference streams over the items keeping a sorted k-array with strict ">" comparisons -- for the new item
// AND for every element it displaces, so a displaced element jumps over its equals: the order inside a group of
// equal similarities (and which members of a group cut by the k-th place survive) depends on the arrival history.
// Everything else is a plain top-k, so the parallel part only has to find the CANDIDATE SET
//   C = { items with similarity >= the k-th best similarity, > 0, not ignored }
// (normally exactly k items), and the reference's own sequential insertion is then replayed over C in item order
// -- items outside C never enter above a member of C, so the replay is bit-identical to the full loop:
//   knn_tile_kernel : one workgroup per (2048-item tile, query): scores into LDS, k rounds of workgroup arg-max
//                     -> the tile's k best by (similarity desc, index asc), + a flag when equal items were cut off
//   knn_merge_kernel: one workgroup per query: k-way merge of the tile lists -> threshold; tiles holding members
//                     of C are replayed in order (a flagged tile is re-scored in full)
// HBM-bound scan (V*D*8 bytes per query); nothing here is GEMM-shaped and nothing is reshaped into one.
#include "common.h"

using namespace goctr;

struct goctr_searcher {
  goctr::Engine* const eng = &goctr::engine();   // the engine (device, streams, arena) the handle was created on
  int64_t V = 0; int D = 0;
  DevBuf<double> items, norms, q, cand_sim, out_sim;
  DevBuf<long long> cand_idx, out_idx, ignore;
  DevBuf<int> out_cnt, cand_cut;
  std::mutex mu;
};

namespace {

constexpr int KNN_TILE = 2048;     // items per workgroup
constexpr int KNN_PER = KNN_TILE / 256;
constexpr int KNN_MAX_K = 256;

__global__ void knn_norm_kernel(const double* items, long long V, int D, double* norms) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= V) return;
  const double* v = items + (size_t)i * D;
  double n = 0;
  for (int d = 0; d < D; ++d) n += v[d] * v[d];
  norms[i] = sqrt(n);
}

// (a, ia) ranks before (b, ib)?  similarity descending, index ascending; idx < 0 = nothing
__device__ __forceinline__ bool knn_before(double a, long long ia, double b, long long ib) {
  if (ib < 0) return ia >= 0;
  if (ia < 0) return false;
  return a > b || (a == b && ia < ib);
}

// workgroup arg-max of (sim, idx) under knn_before; result broadcast to every thread
__device__ __forceinline__ void knn_block_best(double& s, long long& i, double* sh_s, long long* sh_i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double so = __shfl_xor(s, o, 64);
    const long long io = __shfl_xor(i, o, 64);
    if (knn_before(so, io, s, i)) { s = so; i = io; }
  }
  const int wave = threadIdx.x >> 6;
  __syncthreads();                      // sh_* free again
  if ((threadIdx.x & 63) == 0) { sh_s[wave] = s; sh_i[wave] = i; }
  __syncthreads();
  s = sh_s[0]; i = sh_i[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (knn_before(sh_s[w], sh_i[w], s, i)) { s = sh_s[w]; i = sh_i[w]; }
}

__global__ __launch_bounds__(256) void knn_tile_kernel(const double* __restrict__ items, const double* __restrict__ norms,
                                                       long long V, int D, const double* __restrict__ queries,
                                                       const long long* __restrict__ ignore, int k, int ntiles,
                                                       double* cand_sim, long long* cand_idx, int* cand_cut) {
  extern __shared__ __attribute__((aligned(16))) double knn_smem[];
  double* qv = knn_smem;                 // [D]
  double* sc = knn_smem + D;             // [KNN_TILE] scores of the tile (<= 0: not a candidate)
  __shared__ double sh_s[4];
  __shared__ long long sh_i[4];
  const int tile = blockIdx.x, q = blockIdx.y;
  const double* query = queries + (size_t)q * D;
  for (int d = threadIdx.x; d < D; d += 256) qv[d] = query[d];
  __syncthreads();
  // the query norm, like every thread of the reference would compute it (embutil.Norm, search.go:86-90)
  double qn = 0;
  for (int d = 0; d < D; ++d) qn += qv[d] * qv[d];
  qn = sqrt(qn);
  const long long ig = ignore ? ignore[q] : -1;
  const long long base = (long long)tile * KNN_TILE;
#pragma unroll
  for (int j = 0; j < KNN_PER; ++j) {
    const int li = j * 256 + threadIdx.x;
    const long long it = base + li;
    double score = 0;
    if (it < V && it != ig) {
      const double n2 = norms[it];
      if (qn != 0 && n2 != 0) {
        const double* v = items + (size_t)it * D;
        double dot = 0;
        for (int d = 0; d < D; ++d) dot += qv[d] * v[d];
        score = dot / qn / n2;
      }
    }
    sc[li] = score;
  }
  __syncthreads();
  double* os = cand_sim + ((size_t)q * ntiles + tile) * k;
  long long* oi = cand_idx + ((size_t)q * ntiles + tile) * k;
  for (int r = 0; r < k; ++r) {
    double bs = 0; long long bi = -1;
#pragma unroll
    for (int j = 0; j < KNN_PER; ++j) {
      const int li = j * 256 + threadIdx.x;
      const double s = sc[li];
      if (s > 0 && knn_before(s, base + li, bs, bi)) { bs = s; bi = base + li; }
    }
    knn_block_best(bs, bi, sh_s, sh_i);
    if (threadIdx.x == 0) { os[r] = bi >= 0 ? bs : 0.0; oi[r] = bi; }
    if (bi < 0) {                        // tile exhausted: the remaining slots are empty
      for (int rr = r + 1 + threadIdx.x; rr < k; rr += 256) { os[rr] = 0.0; oi[rr] = -1; }
      break;
    }
    if (threadIdx.x == (int)((bi - base) & 255)) sc[bi - base] = 0.0;   // remove the winner
    __syncthreads();
    if (r == k - 1) {                    // list full: were items equal to its last entry left behind?
      int cut = 0;
#pragma unroll
      for (int j = 0; j < KNN_PER; ++j) cut |= sc[j * 256 + threadIdx.x] == bs;
      cut = __syncthreads_or(cut);
      if (threadIdx.x == 0) cand_cut[(size_t)q * ntiles + tile] = cut;
      return;
    }
  }
  if (threadIdx.x == 0) cand_cut[(size_t)q * ntiles + tile] = 0;
}

// score of one item exactly as the tile kernel / the reference compute it
__device__ __forceinline__ double knn_score(const double* items, const double* norms, long long it, int D, const double* qv,
                                            double qn) {
  const double n2 = norms[it];
  if (qn == 0 || n2 == 0) return 0;
  const double* v = items + (size_t)it * D;
  double dot = 0;
  for (int d = 0; d < D; ++d) dot += qv[d] * v[d];
  return dot / qn / n2;
}

// search.go:104-121 for one arriving item (thread 0 only): strict ">" for the item and for whatever it displaces
__device__ __forceinline__ void knn_insert(double* nb_s, long long* nb_i, int k, double score, long long it, double& low) {
  if (!(score > low)) return;
  double ts = score; long long ti = it;
  for (int i = 0; i < k; ++i) {
    if (ts > nb_s[i]) {
      const double xs = nb_s[i]; const long long xi = nb_i[i];
      nb_s[i] = ts; nb_i[i] = ti;
      ts = xs; ti = xi;
    }
  }
  low = nb_s[k - 1];
}

__global__ __launch_bounds__(256) void knn_merge_kernel(const double* __restrict__ items, const double* __restrict__ norms,
                                                        long long V, int D, const double* __restrict__ queries,
                                                        const long long* __restrict__ ignore, const double* cand_sim,
                                                        const long long* cand_idx, const int* cand_cut, int ntiles, int k,
                                                        long long* out_idx, double* out_sim, int* out_cnt) {
  extern __shared__ __attribute__((aligned(16))) double knn_msmem[];
  double* qv = knn_msmem;                       // [D]
  double* sc = knn_msmem + D;                   // [KNN_TILE] re-scored tile
  double* nb_s = sc + KNN_TILE;                 // [k]
  long long* nb_i = reinterpret_cast<long long*>(nb_s + k);   // [k]
  long long* tmp_i = nb_i + k;                  // [k] a tile's candidates, sorted by index
  double* tmp_s = reinterpret_cast<double*>(tmp_i + k);       // [k]
  __shared__ double sh_s[4];
  __shared__ long long sh_i[4];
  __shared__ int sh_t[4];
  __shared__ unsigned involved[8192 / 32];      // tiles holding members of the candidate set
  const int q = blockIdx.x;
  const double* cs = cand_sim + (size_t)q * ntiles * k;
  const long long* ci = cand_idx + (size_t)q * ntiles * k;
  const int* cc = cand_cut + (size_t)q * ntiles;
  for (int d = threadIdx.x; d < D; d += 256) qv[d] = queries[(size_t)q * D + d];
  for (int w = threadIdx.x; w < 8192 / 32; w += 256) involved[w] = 0u;
  for (int r = threadIdx.x; r < k; r += 256) { nb_s[r] = 0.0; nb_i[r] = -1; }
  __syncthreads();
  double qn = 0;
  for (int d = 0; d < D; ++d) qn += qv[d] * qv[d];
  qn = sqrt(qn);
  const long long ig = ignore ? ignore[q] : -1;

  // ---- phase A: k-way merge of the tile lists -> similarity of the k-th best (the threshold of C)
  constexpr int MAXT = 32;                      // tiles per thread: 8192 tiles = 16.7 M items
  int head[MAXT];
#pragma unroll
  for (int u = 0; u < MAXT; ++u) head[u] = 0;
  int filled = 0;
  double last = 0;
  for (int r = 0; r < k; ++r) {
    double bs = 0; long long bi = -1; int bt = -1;
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
      const int t = u * 256 + threadIdx.x;
      if (t < ntiles && head[u] < k) {
        const long long i = ci[(size_t)t * k + head[u]];
        const double s = cs[(size_t)t * k + head[u]];
        if (i >= 0 && knn_before(s, i, bs, bi)) { bs = s; bi = i; bt = t; }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double so = __shfl_xor(bs, o, 64);
      const long long io = __shfl_xor(bi, o, 64);
      const int to = __shfl_xor(bt, o, 64);
      if (knn_before(so, io, bs, bi)) { bs = so; bi = io; bt = to; }
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh_s[wave] = bs; sh_i[wave] = bi; sh_t[wave] = bt; }
    __syncthreads();
    bs = sh_s[0]; bi = sh_i[0]; bt = sh_t[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (knn_before(sh_s[w], sh_i[w], bs, bi)) { bs = sh_s[w]; bi = sh_i[w]; bt = sh_t[w]; }
    if (bi < 0) break;
    ++filled; last = bs;
    if ((bt & 255) == (int)threadIdx.x) {
#pragma unroll
      for (int u = 0; u < MAXT; ++u) if (u == (bt >> 8)) head[u] += 1;
    }
  }
  // C = similarity >= thr (and > 0): the k-th best when k items qualify, else everything positive
  const double thr = filled == k ? last : 0.0;
  // tiles holding members of C: they contributed a winner, or their next unconsumed entry still reaches thr
#pragma unroll
  for (int u = 0; u < MAXT; ++u) {
    const int t = u * 256 + threadIdx.x;
    if (t < ntiles) {
      bool in = head[u] > 0;
      if (!in && head[u] < k) {
        const long long i = ci[(size_t)t * k + head[u]];
        in = i >= 0 && cs[(size_t)t * k + head[u]] >= thr;
      } else if (in && head[u] < k) {
        // (ties right behind the consumed part are found by the replay itself)
      }
      if (in) atomicOr(&involved[t >> 5], 1u << (t & 31));
    }
  }
  __syncthreads();

  // ---- phase B: replay the reference's insertion over C, tile by tile in item order
  double low = 0;
  for (int w = 0; w < (ntiles + 31) / 32; ++w) {
    unsigned bits = involved[w];
    while (bits) {
      const int t = w * 32 + __ffs(bits) - 1;
      bits &= bits - 1;
      const double* ts = cs + (size_t)t * k;
      const long long* ti = ci + (size_t)t * k;
      // a full list that cut off items equal to its last entry, and that entry belongs to C: re-score the tile
      const bool rescan = cc[t] != 0 && ts[k - 1] >= thr && ts[k - 1] > 0;
      if (rescan) {
        const long long base = (long long)t * KNN_TILE;
#pragma unroll
        for (int j = 0; j < KNN_PER; ++j) {
          const int li = j * 256 + threadIdx.x;
          const long long it = base + li;
          sc[li] = (it < V && it != ig) ? knn_score(items, norms, it, D, qv, qn) : 0.0;
        }
        __syncthreads();
        if (threadIdx.x == 0)
          for (int li = 0; li < KNN_TILE; ++li)
            if (sc[li] > 0 && sc[li] >= thr) knn_insert(nb_s, nb_i, k, sc[li], base + li, low);
        __syncthreads();
      } else if (threadIdx.x == 0) {
        int n = 0;
        for (int r = 0; r < k; ++r) {
          if (ti[r] < 0 || !(ts[r] >= thr) || !(ts[r] > 0)) break;        // sorted by similarity: the rest is below thr
          int p = n++;                                                     // insertion sort by item index
          while (p > 0 && tmp_i[p - 1] > ti[r]) { tmp_i[p] = tmp_i[p - 1]; tmp_s[p] = tmp_s[p - 1]; --p; }
          tmp_i[p] = ti[r]; tmp_s[p] = ts[r];
        }
        for (int r = 0; r < n; ++r) knn_insert(nb_s, nb_i, k, tmp_s[r], tmp_i[r], low);
      }
    }
  }
  __syncthreads();
  int cnt = 0;
  for (int r = 0; r < k; ++r) cnt += nb_i[r] >= 0;
  for (int r = threadIdx.x; r < k; r += 256) { out_idx[(size_t)q * k + r] = nb_i[r]; out_sim[(size_t)q * k + r] = nb_s[r]; }
  // search.go:126-131: `if neighbors[i].Word == "" { k = i }` keeps overwriting k => k-1 entries when any is empty
  if (threadIdx.x == 0) out_cnt[q] = cnt < k ? k - 1 : k;
}

}  // namespace

extern "C" {

int goctr_searcher_create(const double* items, int64_t V, int D, goctr_searcher** out) {
  GOCTR_ENTER();
  GOCTR_CHECK(items && out && V > 0 && D > 0, "goctr_searcher_create: bad arguments");
  GOCTR_CHECK(D <= 1024, "goctr_searcher_create: dim %d > 1024", D);
  GOCTR_CHECK(cdiv(V, KNN_TILE) <= 8192, "goctr_searcher_create: more than %d items", 8192 * KNN_TILE);
  goctr_searcher* s = new goctr_searcher;
  s->V = V; s->D = D;
  if (s->items.alloc((size_t)V * D, false) || s->items.upload(items, (size_t)V * D) || s->norms.alloc((size_t)V, false)) {
    delete s;
    return -1;
  }
  hipLaunchKernelGGL(knn_norm_kernel, dim3((unsigned)cdiv(V, 256)), dim3(256), 0, engine().stream, s->items.p, (long long)V, D,
                     s->norms.p);
  if (hipGetLastError() != hipSuccess) { set_error("knn_norm_kernel launch failed"); delete s; return -1; }
  *out = s;
  return 0;
}

void goctr_searcher_destroy(goctr_searcher* s) { delete s; }

int goctr_searcher_search(goctr_searcher* s, const double* queries, int Q, int k, const int64_t* ignore, int64_t* out_idx,
                          double* out_sim, int* out_count) {
  GOCTR_ENTER_H(s);
  GOCTR_CHECK(s && queries && out_idx && out_sim && out_count, "goctr_searcher_search: null argument");
  GOCTR_CHECK(Q > 0 && k > 0 && k <= KNN_MAX_K, "goctr_searcher_search: Q %d, k %d (k <= %d)", Q, k, KNN_MAX_K);
  std::lock_guard<std::mutex> lk(s->mu);
  Engine& e = engine();
  const int ntiles = (int)cdiv(s->V, KNN_TILE);
  if (s->q.ensure((size_t)Q * s->D, false) || s->q.upload(queries, (size_t)Q * s->D)) return -1;
  if (s->ignore.ensure((size_t)Q, false)) return -1;
  {
    std::vector<long long> ig(Q, -1);
    if (ignore) for (int i = 0; i < Q; ++i) ig[i] = ignore[i];
    if (s->ignore.upload(ig.data(), Q)) return -1;
  }
  if (s->cand_sim.ensure((size_t)Q * ntiles * k, false) || s->cand_idx.ensure((size_t)Q * ntiles * k, false) ||
      s->cand_cut.ensure((size_t)Q * ntiles, false)) return -1;
  if (s->out_sim.ensure((size_t)Q * k, false) || s->out_idx.ensure((size_t)Q * k, false) || s->out_cnt.ensure(Q, false)) return -1;
  const size_t lds = sizeof(double) * ((size_t)s->D + KNN_TILE);
  hipLaunchKernelGGL(knn_tile_kernel, dim3(ntiles, Q), dim3(256), lds, e.stream, s->items.p, s->norms.p, (long long)s->V, s->D,
                     s->q.p, s->ignore.p, k, ntiles, s->cand_sim.p, s->cand_idx.p, s->cand_cut.p);
  GOCTR_HIP(hipGetLastError());
  const size_t lds_m = sizeof(double) * ((size_t)s->D + KNN_TILE + 4 * (size_t)k);
  hipLaunchKernelGGL(knn_merge_kernel, dim3(Q), dim3(256), lds_m, e.stream, s->items.p, s->norms.p, (long long)s->V, s->D, s->q.p,
                     s->ignore.p, s->cand_sim.p, s->cand_idx.p, s->cand_cut.p, ntiles, k, s->out_idx.p, s->out_sim.p, s->out_cnt.p);
  GOCTR_HIP(hipGetLastError());
  std::vector<long long> oi((size_t)Q * k);
  if (s->out_idx.download(oi.data(), oi.size()) || s->out_sim.download(out_sim, (size_t)Q * k) || s->out_cnt.download(out_count, Q))
    return -1;
  for (size_t i = 0; i < oi.size(); ++i) out_idx[i] = oi[i];
  return 0;
}

}  // extern "C"
