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
//
// Round 4: the SCAN path (default for D = 16 / 32 / 64, k < 64; GOCTR_KNN_SCAN=0 keeps the two kernels above).  The tile kernel
// reads the item matrix once per QUERY (64 queries per call: 8.2 GB through the MALL for a 128 MB matrix) and pays k
// workgroup-wide arg-max rounds and two float64 divisions per (item, query) to rank 2048 items of which, for almost every tile,
// none ends up among the k neighbours.  The scan path is FILTER + EXACT REFINE:
//   knn_scan_kernel    filter, float32: the items are kept a second time as NORMALISED float32 rows (v / |v|, made once), a call's
//                      queries likewise; one workgroup per tile of 1024 items and block of 64 queries: a thread holds its 4 rows in
//                      registers (each row is read once per call), a query's components arrive by scalar loads as SGPR operands of
//                      v_pk_fma_f32; per (item, query) the approximate cosine a = sum q^_d v^_d, |a - sim| <= E = (D + 8) 2^-23
//                      (rounding of the inputs + D fused multiply-adds, products bounded by Cauchy-Schwarz); the tile's maximum
//                      per query AND the maximum of every sub-block of the tile (32 consecutive items / the rows of 16 adjacent
//                      threads).  From 12 queries per call on the filter runs on the matrix cores: two bf16 planes and three
//                      products (knn_scan_bf16_kernel, |a - sim| <= 2^-14, HBM-bound: every row once per call at ~5 TB/s)
//   knn_collect_kernel ONE workgroup per query: L = the k-th largest of 256 group maxima (the (k + 1)-th when an item is ignored): k
//                      distinct items have a >= L, so the k-th best similarity is >= L - E, every member of the candidate set C has
//                      sim >= L - E and lies in a tile AND a sub-block whose maximum is >= L - 2 E -- those sub-blocks, about k + 3
//                      per query, pass the float32 filter again and the survivors are scored EXACTLY (float64, the reference's
//                      d-order and its two divisions); items with sim >= L - E are the candidates (LDS), which the same workgroup
//                      sorts by item index and replays the reference's insertion over.  Items outside C never stand above a
//                      member of C in the k-array, so replaying any superset of C in item order leaves the same array as the
//                      full loop -- bit-exact results from an approximate filter.  More than 2048 candidates or a bound <= 0
//                      (fewer than k positive group maxima, masses of equal similarities): the call falls back to the tile kernels.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

using namespace goctr;

struct goctr_searcher {
  goctr::Engine* const eng = &goctr::engine();   // the engine (device, streams, arena) the handle was created on
  int64_t V = 0; int D = 0;
  DevBuf<double> items, norms, q, cand_sim, out_sim;
  DevBuf<long long> cand_idx, out_idx, ignore;
  DevBuf<int> out_cnt, cand_cut;
  // scan path: 1 / norm per item, the call's packed input (queries | ignore) and output (idx | sim | count), per-tile maxima,
  // bounds, candidate lists; pinned staging for ONE upload and ONE download per call
  DevBuf<float> items32, tmax;               // normalised float32 rows; per (query, tile) maxima
  DevBuf<unsigned short> items_bf;           // the same rows as two bf16 planes [2][rows padded][D]: hi = bf16(x), lo = bf16(x - hi)
  DevBuf<float> bmax;                        // per (tile, sub-block, query) maxima [tiles][sub-blocks][padded queries]
  DevBuf<unsigned char> in_pack, out_pack;
  void* h_in = nullptr; void* h_out = nullptr; size_t h_in_bytes = 0, h_out_bytes = 0;
  // the call's input in fine-grained DEVICE memory that the host writes through the PCIe BAR (large-BAR systems): no copy command
  // in front of the scan kernel.  bar_state: 0 untried, 1 allocated, -1 refused by the runtime (the staged copy stays)
  unsigned char* in_bar = nullptr; size_t in_bar_bytes = 0; int bar_state = 0;
  std::vector<void*> retired_dev;            // outgrown in_bar buffers (hipFree waits for the whole device: with the handle)
  bool lds_ok = false;
  std::mutex mu;
  std::vector<void*> retired;                // outgrown pinned buffers: freed with the handle (hipHostFree waits for the whole device)
  ~goctr_searcher() { for (void* p : retired_dev) (void)hipFree(p); if (in_bar) (void)hipFree(in_bar); for (void* p : retired) (void)hipHostFree(p); if (h_in) (void)hipHostFree(h_in); if (h_out) (void)hipHostFree(h_out); }
};

namespace {

constexpr int KNN_TILE = 2048;     // items per workgroup
constexpr int KNN_PER = KNN_TILE / 256;
constexpr int KNN_MAX_K = 256;
constexpr int KNN_PENDING = INT32_MIN;   // a query's count in the pinned output while its workgroup has not finished

__global__ void knn_norm_kernel(const double* items, long long V, int D, double* norms, float* items32, unsigned short* items_bf,
                                long long plane_elems) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= V) return;
  const double* v = items + (size_t)i * D;
  double n = 0;
  for (int d = 0; d < D; ++d) n += v[d] * v[d];
  n = sqrt(n);
  norms[i] = n;
  if (items32) {                             // scan path: v / |v| in float32 (a zero-norm item scores 0, as searchutil.go:21-23 returns)
    const double r = n != 0 ? 1.0 / n : 0.0;
    for (int d = 0; d < D; ++d) {
      const float x = (float)(v[d] * r);
      items32[(size_t)i * D + d] = x;
      if (items_bf) {                        // x = hi + lo + rest, |rest| <= 2^-16 |x| (two round-to-nearest bf16 steps)
        const __bf16 h = (__bf16)x;
        const __bf16 lo = (__bf16)(x - (float)h);
        items_bf[(size_t)i * D + d] = __builtin_bit_cast(unsigned short, h);
        items_bf[(size_t)plane_elems + (size_t)i * D + d] = __builtin_bit_cast(unsigned short, lo);
      }
    }
  }
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

// ------------------------------------------------------------------------------------------------------------- scan path
constexpr int KNN2_QB = 64;        // queries per workgroup (the rows are read once per block: once per call up to 64 queries)
constexpr int KNN2_CAP = 2048;     // candidates per query the replay kernel takes

template <int CTRL>
__device__ __forceinline__ float knn_dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// knn_scan_kernel<D, IPT>: a thread keeps IPT normalised float32 rows in registers (read ONCE) and walks the block's queries; a
// query's components are wavefront-uniform and arrive by SCALAR loads (constant address space) straight into the SGPR operands of
// the packed multiply-adds -- no LDS traffic and no vector registers for them.  (Measured on the way, V = 10^6, D = 16, 64 queries
// per call, float64 filter: query fragments as broadcast LDS reads, one item per read: 152 us, LDS-issue-bound at 2.4 x its VALU
// time; two items per read with 256 registers: 219 us; scalar loads but the rows re-read per query group from L2: 575 us; rows in
// registers + scalar loads: 104 us, VALU-bound on 32 float64 multiplies and adds per pair -- which an APPROXIMATE filter does not
// need: float32, 8 packed FMAs per pair.)
template <int D, int IPT>
__global__ __launch_bounds__(256) void knn_scan_kernel(const float* __restrict__ items32, long long V,
                                                       const float* __restrict__ q32 /* normalised, padded to whole blocks */, int Q,
                                                       int nt, float* __restrict__ tmax, float* __restrict__ bmax, int qpad) {
  static_assert(D % 16 == 0, "scalar-load batches of 16 floats");
  constexpr int TILE = 256 * IPT;
  __shared__ float red[16 * KNN2_QB];                  // [16 rows of 16 lanes][QB]
  const int tile = blockIdx.x, q0 = blockIdx.y * KNN2_QB;
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  f2 rowv[IPT][D / 2];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const long long it = (long long)tile * TILE + j * 256 + threadIdx.x;
    const bool in = it < V;
    const float* v = items32 + (size_t)(in ? it : V - 1) * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      const f4 x = *reinterpret_cast<const f4*>(v + d);
      rowv[j][d / 2] = in ? f2{x[0], x[1]} : f2{0.f, 0.f};          // (items past the end score 0)
      rowv[j][d / 2 + 1] = in ? f2{x[2], x[3]} : f2{0.f, 0.f};
    }
  }
  const int row = threadIdx.x >> 4;
  typedef const f2 __attribute__((address_space(4))) cf2;
  cf2* qbase = (cf2*)(unsigned long long)(q32 + (size_t)q0 * D);
  const int nq = Q - q0 < KNN2_QB ? Q - q0 : KNN2_QB;
  // one query: D / 2 packed FMAs per row, the rows' scores and 0 folded with v_max3, the workgroup's maximum by integer maxima
  // over DPP lane patterns (the scores are >= 0 and never NaN there: their bit patterns order like the values -- one
  // instruction per step, where a float maximum of a DPP move costs a move, two canonicalising maxima and the maximum)
  auto one_query = [&](int q, const f2 (&qf)[D / 2]) {
    f2 acc[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) acc[j] = f2{0.f, 0.f};
#pragma unroll
    for (int e = 0; e < D / 2; ++e)
#pragma unroll
      for (int j = 0; j < IPT; ++j) acc[j] = __builtin_elementwise_fma(qf[e], rowv[j][e], acc[j]);
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < IPT; ++j) m = __builtin_fmaxf(m, acc[j][0] + acc[j][1]);     // (a NaN score -- an item with an infinite norm -- is dropped)
    int mi = __builtin_bit_cast(int, m);
    mi = max(mi, __builtin_amdgcn_update_dpp(0, mi, 0xB1, 0xf, 0xf, false));
    mi = max(mi, __builtin_amdgcn_update_dpp(0, mi, 0x4E, 0xf, 0xf, false));
    mi = max(mi, __builtin_amdgcn_update_dpp(0, mi, 0x141, 0xf, 0xf, false));
    mi = max(mi, __builtin_amdgcn_update_dpp(0, mi, 0x140, 0xf, 0xf, false));
    if ((threadIdx.x & 15) == 0) red[row * KNN2_QB + q] = __builtin_bit_cast(float, mi);
  };
  // QG queries' components are requested together (scalar loads return out of order: the only wait is for ALL outstanding ones,
  // so one load per query in front of its FMAs exposes one scalar-cache latency per query; a group exposes one per QG queries).
  // The block's query rows are padded with zero rows to whole blocks, so a group may run past nq.
  constexpr int QG = D == 16 ? 4 : 2;                 // (QG x D / 2 SGPR pairs)
  auto load_query = [&](int q, f2 (&qf)[D / 2]) {
#pragma unroll
    for (int e = 0; e < D / 2; ++e) qf[e] = qbase[(size_t)q * (D / 2) + e];
  };
#pragma unroll 1
  for (int q = 0; q < nq; q += QG) {
    f2 qf[QG][D / 2];
#pragma unroll
    for (int u = 0; u < QG; ++u) load_query(q + u, qf[u]);
#pragma unroll
    for (int u = 0; u < QG; ++u) one_query(q + u, qf[u]);
  }
  __syncthreads();
  if ((int)threadIdx.x < nq) {
    // sub-block r of the tile = the rows of threads 16 r .. 16 r + 15 (items j 256 + 16 r + i): its maximum goes to
    // bmax [tile][16][queries] (the collect pass reads only the sub-blocks that can hold a candidate), the tile's to tmax
    float m = 0.f;
    for (int r = 0; r < 16; ++r) {
      const float x = red[r * KNN2_QB + threadIdx.x];
      bmax[((size_t)tile * 16 + r) * qpad + q0 + threadIdx.x] = x;
      m = fmaxf(m, x);
    }
    tmax[(size_t)(q0 + threadIdx.x) * nt + tile] = m;
  }
}

// knn_scan_bf16_kernel<D>: the filter on the bf16 matrix cores (round 5) -- the block's scores are a [1024 items] x [64 queries]
// product over D dimensions; a wavefront takes 32 items at a time as the rows of a 32 x 32 tile, the queries as its columns; a lane
// then holds 16 items' scores of ONE query, folds them into the sub-block's and its running maximum, and the lanes / wavefronts of a
// workgroup are combined once at the end.  (Round 4's float32-input MFMA kernel, v_mfma_f32_32x32x2_f32, issued at a sixteenth of the
// bf16 rate and was what bound the scan: MFMA-busy 60 %, 2.5 TB/s of rows = 0.32 of HBM, 26 us; it left the tree in round 5,
// profiles/r04_knn_scan_ab.txt and git keep it.)  The normalised rows are padded with zero rows to whole tiles: no tail checks.  Rows and queries are
// kept as TWO bf16 planes, x = hi + lo + rest with |rest| <= 2^-16 |x|, and a score is hi.hi + hi.lo + lo.hi on
// v_mfma_f32_32x32x16_bf16 (products of bf16 pairs are exact in float32; the dropped lo.lo and the two rests are <= 3 x 2^-16 of
// sum |q_d v_d| <= 1): |a - sim| <= E' = 2^-14 with room for the float32 accumulation (tests/test_knn_filter_bound.py) -- twenty
// times the float32 filter's E, still two orders below the spacing of the best similarities of a 10^6-item catalogue, so the
// candidate sets hardly grow.  One MFMA step covers k = 16: at D = 16 a 32-item x 32-query block is THREE instructions.
template <int D>
__global__ __launch_bounds__(256) void knn_scan_bf16_kernel(const unsigned short* __restrict__ items_bf, long long plane_elems,
                                                            const unsigned short* __restrict__ q_bf /* [2][padded Q][D] */, long long qplane,
                                                            int Q, int nt, float* __restrict__ tmax, float* __restrict__ bmax, int qpad) {
  // (round 6 measured these maxima stored THROUGH the L2 -- sc1, as the weight-gradient slabs are now: no effect on the call,
  // 9.80 vs 9.85 ms per 200 calls; profiles/r06_write_through.txt)
  typedef float v16 __attribute__((ext_vector_type(16)));
  typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  constexpr int KS = D / 16;                           // MFMA k-steps per row
  __shared__ float red[4][KNN2_QB];
  const int tile = blockIdx.x, q0 = blockIdx.y * KNN2_QB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, kg = lane >> 5;
  // B operands: query q0 + 32 h + col, components 16 s + 8 kg .. + 8, both planes
  bf8 qh[2][KS], ql[2][KS];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int sx = 0; sx < KS; ++sx) {
      const size_t o = (size_t)(q0 + h * 32 + col) * D + 16 * sx + 8 * kg;
      qh[h][sx] = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(q_bf + o));
      ql[h][sx] = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(q_bf + qplane + o));
    }
  float mx[2] = {0.f, 0.f};
  const size_t row0 = (size_t)tile * 1024 + (size_t)wave * 256;
#pragma unroll 2
  for (int t = 0; t < 8; ++t) {                        // 8 blocks of 32 items per wavefront; a lane loads 16 bytes per plane and k-step
    bf8 ah[KS], al[KS];
#pragma unroll
    for (int sx = 0; sx < KS; ++sx) {
      const size_t o = (row0 + (size_t)t * 32 + col) * D + 16 * sx + 8 * kg;
      ah[sx] = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(items_bf + o));
      al[sx] = __builtin_bit_cast(bf8, *reinterpret_cast<const u4*>(items_bf + plane_elems + o));
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      v16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int sx = 0; sx < KS; ++sx) {                // (small terms first)
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[sx], qh[h][sx], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[sx], ql[h][sx], c, 0, 0, 0);
      }
#pragma unroll
      for (int sx = 0; sx < KS; ++sx) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[sx], qh[h][sx], c, 0, 0, 0);
      float m = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) m = __builtin_fmaxf(m, c[r]);         // (a NaN score -- an item with an infinite norm -- is dropped)
      // this 32-item block's maximum per query (both row halves): bmax [tile][32 blocks][queries]
      m = __builtin_fmaxf(m, __shfl_xor(m, 32, 64));
      if (kg == 0) bmax[((size_t)tile * 32 + wave * 8 + t) * qpad + q0 + h * 32 + col] = m;
      mx[h] = __builtin_fmaxf(mx[h], m);
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float m = mx[h];
    m = __builtin_fmaxf(m, __shfl_xor(m, 32, 64));     // the other 16 rows of the column
    if (kg == 0) red[wave][h * 32 + col] = m;
  }
  __syncthreads();
  if (threadIdx.x < KNN2_QB && q0 + (int)threadIdx.x < Q) {
    const float m = __builtin_fmaxf(__builtin_fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), __builtin_fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
    tmax[(size_t)(q0 + threadIdx.x) * nt + tile] = m;
  }
}

// DPP lane exchange (quad_perm / row_half_mirror / row_mirror), as ctr_kernels.h dpp_f32
template <int CTRL>
__device__ __forceinline__ float dpp_f32k(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// 64-bit v_readlane (the lane index must be wave-uniform)
__device__ __forceinline__ long long knn_readlane(long long x, int lane) {
  const int lo = __builtin_amdgcn_readlane((int)(unsigned int)(unsigned long long)x, lane);
  const int hi = __builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)x >> 32), lane);
  return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo);
}
__device__ __forceinline__ double knn_readlane(double x, int lane) {
  return __builtin_bit_cast(double, knn_readlane(__builtin_bit_cast(long long, x), lane));
}

// DPP wave_shr:1 (gfx9): lane l gets lane l - 1's value, lane 0 its own -- a pure vector move, no LDS crossbar
__device__ __forceinline__ int knn_wave_shr1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ long long knn_wave_shr1(long long x) {
  const int lo = knn_wave_shr1((int)(unsigned int)(unsigned long long)x), hi = knn_wave_shr1((int)(unsigned int)((unsigned long long)x >> 32));
  return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo);
}
__device__ __forceinline__ double knn_wave_shr1(double x) {
  return __builtin_bit_cast(double, knn_wave_shr1(__builtin_bit_cast(long long, x)));
}

// knn_collect_kernel: ONE workgroup per query does everything behind the scan -- bound, tile list, sub-block list, float32 filter
// of the listed sub-blocks' items, exact similarities, the reference's insertion replayed over the candidates, results into the
// pinned output buffer.  (Round 4: eight workgroups per query re-read every listed TILE -- 64 KB each, ~45 of them per query, the
// launch at 83 % waiting -- and a third launch replayed; round 5 first folded the replay into the last workgroup to finish --
// no faster, the dependent chain is the same -- and then cut the re-read itself: the scan kernels also leave the maximum of
// every 32-item SUB-BLOCK, so a query re-reads ~13 sub-blocks of 2 KB instead of ~13 tiles of 64 KB, which one workgroup does in
// two rounds of loads, with the candidates in LDS: no global candidate lists, no atomics on them, no fan-in.)
//   bound   L = the `rounds`-th largest (k, + 1 when an item is ignored: it may own one) of 256 GROUP maxima (group = tile mod 256;
//           the maxima belong to distinct items, so k items have a score >= L): every member of the candidate set C has
//           similarity >= L - E and lies in a tile / sub-block whose maximum is >= L - 2 E.  Every thread ranks its own group's
//           maximum among the 256 (broadcast LDS reads; ties by group index: exactly one thread holds the rank).
//   lists   tiles with maximum >= L - 2 E (> 0), then their sub-blocks with maximum >= L - 2 E: compacted into LDS
//   filter  the listed sub-blocks' items pass the float32 filter (its error is inside E whichever scan kernel made the maxima);
//           survivors are scored EXACTLY (float64, the reference's d-order and its two divisions) and join the candidates when they
//           reach L - E
//   replay  the candidates sorted by item index, search.go:104-121 over them by one wavefront with the k-array in registers: an
//           insertion into a non-increasing array puts the item at p = the first position with score > a[p]; the element it
//           displaces is carried down PAST its equals (the reference's strict `>` for displaced elements too) and lands behind its
//           tie run, displacing the first element of the next run, and so on; the last carry is dropped.  So: position p takes the
//           item, every later run START takes the FIRST element of the run before it, everything else stays (quirk Q22;
//           tests/test_knn_replay_rule.py checks the rule against the literal loop).  Two ballots, one shuffle per insertion.
// out_cnt[q] = -1: the query is handed to the exact tile kernels (bound <= 0, or a list outgrew its LDS array).
// sb_mode 0: sub-block b of a tile = items 32 b .. 32 b + 31 (the matrix-core scan kernels); 1: items j 256 + 16 b + i, j < tile / 256,
// i < 16 (the VALU scan kernel: the rows of 16 adjacent threads)
constexpr int KNN2_MYT = 256;      // listed tiles per query
constexpr int KNN2_MYB = 1024;     // listed sub-blocks per query
__global__ __launch_bounds__(256) void knn_collect_kernel(const double* __restrict__ items, const double* __restrict__ norms,
                                                          const float* __restrict__ items32, long long V, int D,
                                                          const double* __restrict__ queries, const float* __restrict__ q32,
                                                          const long long* __restrict__ ignore, const float* __restrict__ tmax,
                                                          const float* __restrict__ bmax, int qpad, int nt, int tile_items, int sb_mode,
                                                          int k, float E, long long* out_idx, double* out_sim, int* out_cnt, int host_polls,
                                                          unsigned long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) double knn_cq[];       // [D] the query | its norm | [D] floats: normalised | the replay's arrays
  float* cq32 = reinterpret_cast<float*>(knn_cq + D + 1);
  double* s_sim = knn_cq + D + 1 + (D + 1) / 2;                          // [CAP] candidates' similarities
  long long* s_idx = reinterpret_cast<long long*>(s_sim + KNN2_CAP);     // [CAP] ... item indices
  double* nb_s = reinterpret_cast<double*>(s_idx + KNN2_CAP);            // [k]
  long long* nb_i = reinterpret_cast<long long*>(nb_s + k);              // [k]
  __shared__ float gmax[256];
  __shared__ float sh_tb, sh_sb;
  __shared__ int wave_cnt[4];
  __shared__ int my_tiles[KNN2_MYT];
  __shared__ int my_blocks[KNN2_MYB];
  __shared__ int n_blk, n_cand;
  const int q = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};           // GOCTR_DBG=knn: phase stamps of workgroup 0, thread 0
  auto stamp = [&](int i) { if (dbg && q == 0 && threadIdx.x == 0) ts[i] = __builtin_amdgcn_s_memtime(); };
  stamp(0);
  const float* tm = tmax + (size_t)q * nt;
  if (threadIdx.x == 0) { n_blk = 0; n_cand = 0; }
  // One round of requests at the launch's cold start: the ignore index and the query's elements (into registers), the tile maxima, and only
  // then the query's copy into LDS and the maximum.  (In the order "copy the query; maxima" the LDS writes waited for the query before the
  // maxima were even asked for: two dependent round trips where one does, ~1.9 k of the launch's ~29 k ticks.)
  long long ig_v = ignore[q];
  const int d0 = (int)threadIdx.x;
  double q64v = 0.0; float q32v = 0.f;
  if (d0 < D) { q64v = queries[(size_t)q * D + d0]; q32v = q32[(size_t)q * D + d0]; }
  // (round 6: the first KNN2_TMR x 256 tile maxima stay in registers for the tile list below -- nt <= 1024 at V = 10^6: no re-read)
  constexpr int KNN2_TMR = 4;
  float tmv[KNN2_TMR];
#pragma unroll
  for (int u = 0; u < KNN2_TMR; ++u) {
    const int t = u * 256 + (int)threadIdx.x;
    tmv[u] = t < nt ? tm[t] : 0.f;
  }
  asm volatile("" : "+v"(ig_v));       // (pinned with this round: left alone, the load sinks to its use behind the maxima's wait)
  const long long ig = ig_v;
  if (d0 < D) { knn_cq[d0] = q64v; cq32[d0] = q32v; }
  for (int d = 256 + threadIdx.x; d < D; d += 256) { knn_cq[d] = queries[(size_t)q * D + d]; cq32[d] = q32[(size_t)q * D + d]; }
  float m = 0.f;
#pragma unroll
  for (int u = 0; u < KNN2_TMR; ++u) m = fmaxf(m, tmv[u]);
  for (int t = KNN2_TMR * 256 + threadIdx.x; t < nt; t += 256) m = fmaxf(m, tm[t]);
  {
    // The bound: the `rounds`-th largest of the 256 group maxima, duplicates counted.  Round 5 ranked every maximum against all 256
    // (256 broadcast LDS reads and ~5 instructions each per thread: 12.6 k of the launch's 43 k cycles, GOCTR_DBG=knn).  Now every
    // wavefront peels off its own `rounds` largest -- a DPP maximum over the rows, four v_readlane, the first lane that holds the
    // maximum drops out -- and the 4 x rounds survivors are ranked among themselves: the same value, ~2 k cycles.
    const int rounds = k + (ig >= 0 ? 1 : 0);                          // <= 64 (knn_scan_usable)
    float mine = m;
    for (int r = 0; r < rounds; ++r) {
      float v = mine;
      v = fmaxf(v, dpp_f32k<0xB1>(v)); v = fmaxf(v, dpp_f32k<0x4E>(v)); v = fmaxf(v, dpp_f32k<0x141>(v)); v = fmaxf(v, dpp_f32k<0x140>(v));
      const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
      const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
      const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
      const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
      const float wmax = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
      const unsigned long long hold = __ballot(mine == wmax);
      if (lane == (int)__builtin_ctzll(hold | (1ull << 63))) mine = -1.f;     // (maxima are >= 0: -1 = peeled off)
      if (lane == 0) gmax[wave * rounds + r] = wmax;                          // (the four lists back to back: 4 x rounds <= 256)
    }
    if (threadIdx.x == 64) {
      double qn = 0;
      for (int d = 0; d < D; ++d) qn += knn_cq[d] * knn_cq[d];           // embutil.Norm (search.go:86-90)
      knn_cq[D] = sqrt(qn);
    }
    __syncthreads();
    stamp(1);
    // (ranking the survivors over v_readlane broadcasts in ONE wavefront instead measured 4.2 k against 2.7 k cycles for this phase)
    if ((int)threadIdx.x < 4 * rounds) {
      const float c = gmax[threadIdx.x];
      int rank = 0;
      for (int u = 0; u < 4 * rounds; ++u) {
        const float o = gmax[u];
        rank += (o > c || (o == c && u < (int)threadIdx.x)) ? 1 : 0;
      }
      if (rank == rounds - 1) { const float kth = c > 0.f ? c : 0.f; sh_tb = kth - 2.f * E; sh_sb = kth - E; }
    }
  }
  __syncthreads();
  stamp(2);
  const float tb = sh_tb;
  const double qn = knn_cq[D];
  const double bd = (double)sh_sb;
  auto give_up = [&]() { if (threadIdx.x == 0) __hip_atomic_store(out_cnt + q, -1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); };
  // kth - 2E <= 0 (fewer than `rounds` positive group maxima, or a k-th maximum inside the filter's error): the candidate set
  // reaches down to every similarity > 0, and an item whose exact similarity lies in (0, E] can have a filter score <= 0 in a
  // tile whose maximum is <= 0 -- the lists below (maximum > 0) would never visit it while the reference returns it (search.go:104:
  // score > low, low = 0): the exact tile kernels take the query.  (A zero query has no neighbours at all -- searchutil.go:21-23 --
  // and goes straight to the replay of nothing.)
  const bool nothing = !(tb > 0.f);
  if (nothing && qn != 0) { give_up(); return; }
  if (!nothing) {
    // ---- listed tiles, in tile order
    int base = 0;
    {
      // the register-cached maxima: every (pass, wavefront) count first, ONE barrier, then the positions (round 5: two barriers per pass)
      __shared__ int pass_cnt[KNN2_TMR][4];
      unsigned long long bal[KNN2_TMR]; bool onv[KNN2_TMR];
#pragma unroll
      for (int u = 0; u < KNN2_TMR; ++u) {
        onv[u] = u * 256 + (int)threadIdx.x < nt && tmv[u] > 0.f && tmv[u] >= tb;
        bal[u] = __ballot(onv[u]);
        if (lane == 0) pass_cnt[u][wave] = __popcll(bal[u]);
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < KNN2_TMR; ++u) {
        int before = base;
        for (int w = 0; w < wave; ++w) before += pass_cnt[u][w];
        const int pos = before + __popcll(bal[u] & ((1ull << lane) - 1ull));
        if (onv[u] && pos < KNN2_MYT) my_tiles[pos] = u * 256 + (int)threadIdx.x;
        base += pass_cnt[u][0] + pass_cnt[u][1] + pass_cnt[u][2] + pass_cnt[u][3];
      }
      __syncthreads();
    }
    for (int t0 = KNN2_TMR * 256; t0 < nt; t0 += 256) {
      const int t = t0 + threadIdx.x;
      const bool on = t < nt && tm[t] > 0.f && tm[t] >= tb;
      const unsigned long long bal = __ballot(on);
      if (lane == 0) wave_cnt[wave] = __popcll(bal);
      __syncthreads();
      int before = base;
      for (int w = 0; w < wave; ++w) before += wave_cnt[w];
      const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
      if (on && pos < KNN2_MYT) my_tiles[pos] = t;
      base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      __syncthreads();
    }
    if (base > KNN2_MYT) { give_up(); return; }               // (uniform)
    stamp(3);
    // ---- their sub-blocks whose own maximum reaches the bound
    const int sb_items = sb_mode == 0 ? 32 : 16 * (tile_items / 256);
    const int SB = tile_items / sb_items;
    for (int i = threadIdx.x; i < base * SB; i += 256) {
      const int tile = my_tiles[i / SB], b = i % SB;
      const float bm = bmax[((size_t)tile * SB + b) * qpad + q];
      if (bm > 0.f && bm >= tb) {
        const int pos = atomicAdd(&n_blk, 1);
        if (pos < KNN2_MYB) my_blocks[pos] = tile * SB + b;
      }
    }
    __syncthreads();
    stamp(4);
    const int nblk = n_blk;
    if (nblk > KNN2_MYB) { give_up(); return; }
    // ---- float32 filter over the listed sub-blocks' items, exact similarity of the survivors
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    auto item_of = [&](int wi) -> long long {
      const int blk = my_blocks[wi / sb_items], e = wi % sb_items;
      const int tile = blk / SB, b = blk - tile * SB;
      const long long it = sb_mode == 0 ? (long long)tile * tile_items + 32 * b + e
                                        : (long long)tile * tile_items + (e >> 4) * 256 + 16 * b + (e & 15);
      return (it < V && it != ig) ? it : -1;
    };
    auto exact = [&](long long it) {
      // the norm and the whole float64 row are requested together
      const double* v = items + (size_t)it * D;
      const double n2 = norms[it];
      double dot = 0;
      if (D == 16) {
        d2 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const d2*>(v + 2 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) { dot += knn_cq[2 * u] * x[u][0]; dot += knn_cq[2 * u + 1] * x[u][1]; }   // d ascending (searchutil.go:17-20)
      } else {
        for (int d = 0; d < D; d += 2) {
          const d2 x = *reinterpret_cast<const d2*>(v + d);
          dot += knn_cq[d] * x[0];
          dot += knn_cq[d + 1] * x[1];
        }
      }
      // (the product pinned before the norm is looked at: the compiler had sunk the row's loads and the 16 multiply-adds behind the
      // "n2 == 0" return -- norm, wait, row, wait: two dependent round trips to cold lines per survivor where the comment above says one)
      asm volatile("" : "+v"(dot));
      if (qn == 0 || n2 == 0) return;
      const double sim = dot / qn / n2;                 // searchutil.go:24-25
      if (!(sim > 0 && sim >= bd)) return;
      const int pos = atomicAdd(&n_cand, 1);
      if (pos < KNN2_CAP) { s_sim[pos] = sim; s_idx[pos] = it; }
    };
    const int nwork = nblk * sb_items;
    if (D == 16) {
      // four items per thread at a time, all 16 row loads in flight before the first FMA
      for (int w0 = threadIdx.x; w0 < nwork; w0 += 4 * 256) {
        f4 x[4][4]; long long its[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int wi = w0 + u * 256;
          its[u] = wi < nwork ? item_of(wi) : -1;
          const float* v32 = items32 + (size_t)(its[u] >= 0 ? its[u] : 0) * 16;
#pragma unroll
          for (int c = 0; c < 4; ++c) x[u][c] = *reinterpret_cast<const f4*>(v32 + 4 * c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (its[u] < 0) continue;
          f2 acc = f2{0.f, 0.f};                         // (knn_scan_kernel's arithmetic: one packed accumulator, d ascending)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            acc = __builtin_elementwise_fma(f2{cq32[4 * c], cq32[4 * c + 1]}, f2{x[u][c][0], x[u][c][1]}, acc);
            acc = __builtin_elementwise_fma(f2{cq32[4 * c + 2], cq32[4 * c + 3]}, f2{x[u][c][2], x[u][c][3]}, acc);
          }
          if (acc[0] + acc[1] >= tb) exact(its[u]);
        }
      }
    } else {
      for (int wi = threadIdx.x; wi < nwork; wi += 256) {
        const long long it = item_of(wi);
        if (it < 0) continue;
        const float* v32 = items32 + (size_t)it * D;
        f2 acc = f2{0.f, 0.f};
        for (int d = 0; d < D; d += 4) {
          const f4 x = *reinterpret_cast<const f4*>(v32 + d);
          acc = __builtin_elementwise_fma(f2{cq32[d], cq32[d + 1]}, f2{x[0], x[1]}, acc);
          acc = __builtin_elementwise_fma(f2{cq32[d + 2], cq32[d + 3]}, f2{x[2], x[3]}, acc);
        }
        if (acc[0] + acc[1] >= tb) exact(it);
      }
    }
  }
  __syncthreads();
  stamp(5);
  const int n = n_cand;
  if (n > KNN2_CAP) { give_up(); return; }
  // ---- replay: sort by item index (an item appears once): own entries to registers, ranks from LDS, write back in order
  for (int r = threadIdx.x; r < k; r += 256) { nb_s[r] = 0.0; nb_i[r] = -1; }
  if (n <= 64) {
    // (round 6: a few dozen candidates at most -- the usual case is k + a handful -- are ordered by ONE wavefront, which then replays
    // them: no workgroup barriers around the sort; the other wavefronts wait at the barrier behind the replay)
    if (threadIdx.x < 64) {
      const long long mi = lane < n ? s_idx[lane] : 0; const double msim = lane < n ? s_sim[lane] : 0.0;
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += knn_readlane(mi, j) < mi ? 1 : 0;
      if (lane < n) { s_idx[rank] = mi; s_sim[rank] = msim; }
    }
  } else {
  long long me[KNN2_CAP / 256]; double ms[KNN2_CAP / 256]; int rk[KNN2_CAP / 256];
#pragma unroll
  for (int u = 0; u < KNN2_CAP / 256; ++u) {
    const int e = threadIdx.x + 256 * u;
    me[u] = 0; ms[u] = 0.0; rk[u] = -1;
    if (e < n) {
      me[u] = s_idx[e]; ms[u] = s_sim[e];
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += s_idx[j] < me[u];
      rk[u] = rank;
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < KNN2_CAP / 256; ++u)
    if (rk[u] >= 0) { s_idx[rk[u]] = me[u]; s_sim[rk[u]] = ms[u]; }
  __syncthreads();
  }
  if (threadIdx.x < 64) {
    double s = 0.0; long long id = -1; double low = 0.0;
    for (int r0 = 0; r0 < n; r0 += 64) {
      const int mm = n - r0 < 64 ? n - r0 : 64;
      const double bs = lane < mm ? s_sim[r0 + lane] : 0.0;
      const long long bi = lane < mm ? s_idx[r0 + lane] : -1;
      for (int j = 0; j < mm; ++j) {
        const double ts = knn_readlane(bs, j);                     // (j, p, k - 1 are wave-uniform: v_readlane, no LDS round trip)
        if (!(ts > low)) continue;                                  // (uniform) search.go:104
        const long long ti = knn_readlane(bi, j);
        const unsigned long long gt = __ballot(lane < k && ts > s);
        if (!gt) continue;
        const int p = __builtin_ctzll(gt);
        const double sprev = knn_wave_shr1(s);                     // (DPP wave_shr:1: no LDS round trip)
        // Round 6: no run of EQUAL similarities among the real entries behind p (empty slots are all (0.0, -1): shifting them changes
        // nothing) => the rule below is a plain shift by one -- every lane behind p takes its predecessor, which the DPP move above
        // already delivered; the general path's two ballots and two LDS shuffles are only for tie groups (quirk Q22).
        if (!__ballot(lane < k && lane > p && sprev == s && s > 0.0)) {
          const long long idprev = knn_wave_shr1(id);
          if (lane == p) { s = ts; id = ti; }
          else if (lane > p && lane < k) { s = sprev; id = idprev; }
          low = knn_readlane(s, k - 1);
          continue;
        }
        const bool start = lane < k && (lane == p || (lane > p && sprev != s));
        const unsigned long long S = __ballot(start);
        const unsigned long long below = lane > 0 ? S & ((1ull << lane) - 1ull) : 0ull;
        const int src = below ? 63 - __builtin_clzll(below) : 0;
        const double fs = __shfl(s, src, 64);
        const long long fid = __shfl(id, src, 64);
        if (lane == p) { s = ts; id = ti; }
        else if (start) { s = fs; id = fid; }
        low = knn_readlane(s, k - 1);
      }
    }
    if (lane < k) { nb_s[lane] = s; nb_i[lane] = id; }
  }
  __syncthreads();
  stamp(6);
  int cnt = 0;
  for (int r = 0; r < k; ++r) cnt += nb_i[r] >= 0;
  const int n_out = cnt < k ? k - 1 : k;                      // search.go:126-131 (see knn_merge_kernel)
  if (host_polls) {
    // the host watches out_cnt[q] in the pinned buffer (knn_search_scan): the neighbours must be there before the count is.
    // Round 6: the neighbours leave as system-scope stores (written through when issued, nothing left dirty), each writer waits for
    // its stores' acknowledgements, and the count follows behind the barrier -- no release fence, whose write-back walk of the
    // XCD's L2 was most of this phase's 4.3 k cycles (round 5: profiles/r05_knn_poll.txt).
    for (int r = threadIdx.x; r < k; r += 256) {
      __hip_atomic_store(out_idx + (size_t)q * k + r, nb_i[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(out_sim + (size_t)q * k + r, nb_s[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(out_cnt + q, n_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    for (int r = threadIdx.x; r < k; r += 256) { out_idx[(size_t)q * k + r] = nb_i[r]; out_sim[(size_t)q * k + r] = nb_s[r]; }
    if (threadIdx.x == 0) out_cnt[q] = n_out;
  }
  stamp(7);
  if (dbg && q == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dbg[i] = ts[i];
    dbg[8] = (unsigned long long)n; dbg[9] = (unsigned long long)n_blk;
  }
}

}  // namespace

// items per workgroup of the scan kernel for this dimension (0: no scan kernel -- the tile kernels serve the call)
// (D = 16: 8 rows per thread measured the same scan time, 41.7 vs 43.6 us, and a slower collect pass: the tiles are twice as large)
static int knn_scan_ipt(int D) { return D == 16 ? 4 : D == 32 ? 4 : D == 64 ? 2 : 0; }
static int knn_scan_tile(int D) { return 256 * knn_scan_ipt(D); }
static int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
// The matrix-core scan kernel (knn_scan_bf16_kernel: D = 16 / 32, where the bf16 planes exist): its cost is per block of 64 query
// columns (13.7 us), the VALU kernel's per query (13 us for one, 34 us for 64).  Measured cross-over between 8 and 16 queries per call
// (profiles/r05_knn_scan_threshold.txt: 8 queries 44.4 vs 45.0 us, 16 46.8 vs 47.2, 24 49.9 vs 47.6, 32 53.6 vs 48.9, 47 63.6 vs
// 51.2): from 12 queries on.  (The 48 of round 4 belonged to the float32 MFMA kernel.)  GOCTR_KNN_MFMA=0 / 1 forces either.
static bool knn_scan_mfma(const goctr_searcher* s, int Q) {
  const char* v = getenv("GOCTR_KNN_MFMA");
  return (v && *v ? *v != '0' : Q >= 12) && knn_scan_tile(s->D) == 1024 && s->items_bf.p && (s->D == 16 || s->D == 32);
}
static bool knn_scan_usable(const goctr_searcher* s, int k) {
  const char* v = getenv("GOCTR_KNN_SCAN");
  if (v && *v == '0') return false;
  const int tile = knn_scan_tile(s->D);
  return tile > 0 && s->items32.p && k + 1 <= 64 && cdiv(s->V, tile) <= (1 << 20);
}
// 0 = done, -1 = error, 1 = fall back to the tile kernels
static int knn_search_scan(goctr_searcher* s, const double* queries, int Q, int k, const int64_t* ignore, int64_t* out_idx,
                           double* out_sim, int* out_count) {
  Engine& e = engine();
  const int D = s->D, tile_items = knn_scan_tile(D), nt = (int)cdiv(s->V, tile_items), nqb = (int)cdiv(Q, KNN2_QB);
  const size_t lds_r = sizeof(double) * (2 * (size_t)KNN2_CAP + 2 * (size_t)k);
  if (!s->lds_ok) {
    GOCTR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(knn_collect_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(double) * (2 * (size_t)KNN2_CAP + 2 * (size_t)KNN_MAX_K + 1024 + 2))));
    s->lds_ok = true;
  }
  // one upload: [Q x D queries | Q ignore | the queries normalised, float32, padded with zero rows to whole 64-query blocks],
  // one download: [Q x k idx | Q x k sim | Q count]
  const size_t in_q = sizeof(double) * (size_t)Q * D, in_ig = sizeof(long long) * (size_t)Q;
  const size_t in_q32 = sizeof(float) * (size_t)nqb * KNN2_QB * D;
  const bool bf = knn_scan_mfma(s, Q);               // the bf16-plane matrix-core filter (from 12 queries per call on, D = 16 / 32)
  const size_t in_qbf = bf ? sizeof(unsigned short) * 2 * (size_t)nqb * KNN2_QB * D : 0, in_bytes = in_q + in_ig + in_q32 + in_qbf;
  const size_t o_idx = sizeof(long long) * (size_t)Q * k, o_sim = sizeof(double) * (size_t)Q * k, out_bytes = o_idx + o_sim + sizeof(int) * (size_t)Q;
  if (s->h_in_bytes < in_bytes) {
    if (s->h_in) s->retired.push_back(s->h_in);
    s->h_in = nullptr; s->h_in_bytes = 0;
    GOCTR_HIP(hipHostMalloc(&s->h_in, in_bytes * 2, hipHostMallocDefault));
    s->h_in_bytes = in_bytes * 2;
  }
  if (s->h_out_bytes < out_bytes) {
    if (s->h_out) s->retired.push_back(s->h_out);
    s->h_out = nullptr; s->h_out_bytes = 0;
    GOCTR_HIP(hipHostMalloc(&s->h_out, out_bytes * 2, hipHostMallocDefault));
    s->h_out_bytes = out_bytes * 2;
  }
  const int qpad = nqb * KNN2_QB;
  // Where the kernels read the input from: device memory the host wrote through the BAR (a 14 KB store burst, ~3 us, instead of a
  // 4.4 us blit kernel on the stream in front of the scan: profiles/r05_knn_bar_input.txt), else the arena buffer behind a copy.
  const char* bv = getenv("GOCTR_KNN_BAR");           // (0: the staged copy, for A/B runs and the tests' comparison)
  const bool want_bar = e.large_bar && s->bar_state >= 0 && !(bv && atoi(bv) == 0);
  if (want_bar && s->in_bar_bytes < in_bytes) {
    if (s->in_bar) s->retired_dev.push_back(s->in_bar);
    s->in_bar = nullptr; s->in_bar_bytes = 0;
    s->in_bar = static_cast<unsigned char*>(bar_alloc(in_bytes * 2));
    if (!s->in_bar) s->bar_state = -1;                 // not on this system: the staged copy from now on
    else { s->in_bar_bytes = in_bytes * 2; s->bar_state = 1; }
  }
  const bool bar = want_bar && s->in_bar != nullptr;
  if ((!bar && s->in_pack.ensure(in_bytes, false)) || s->tmax.ensure((size_t)Q * nt, false) || s->bmax.ensure((size_t)nt * 32 * qpad, false)) return -1;
  unsigned char* const d_in = bar ? s->in_bar : s->in_pack.p;
  memcpy(s->h_in, queries, in_q);
  long long* h_ig = reinterpret_cast<long long*>(static_cast<char*>(s->h_in) + in_q);
  for (int i = 0; i < Q; ++i) h_ig[i] = ignore ? (long long)ignore[i] : -1;
  float* h_q32 = reinterpret_cast<float*>(static_cast<char*>(s->h_in) + in_q + in_ig);
  memset(h_q32, 0, in_q32);
  for (int i = 0; i < Q; ++i) {
    double qn = 0;
    for (int d = 0; d < D; ++d) qn += queries[(size_t)i * D + d] * queries[(size_t)i * D + d];
    qn = std::sqrt(qn);
    if (qn != 0 && std::isfinite(qn))
      for (int d = 0; d < D; ++d) h_q32[(size_t)i * D + d] = (float)(queries[(size_t)i * D + d] / qn);
  }
  if (bf) {                                          // the normalised queries as two bf16 planes (round to nearest even, like the device's casts)
    unsigned short* h_qbf = reinterpret_cast<unsigned short*>(static_cast<char*>(s->h_in) + in_q + in_ig + in_q32);
    const size_t plane = (size_t)nqb * KNN2_QB * D;
    auto f2bf = [](float x) -> unsigned short {
      uint32_t u; memcpy(&u, &x, 4);
      if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
      u += 0x7fffu + ((u >> 16) & 1u);
      return (unsigned short)(u >> 16);
    };
    auto bf2f = [](unsigned short h) -> float { uint32_t u = (uint32_t)h << 16; float x; memcpy(&x, &u, 4); return x; };
    for (size_t i = 0; i < plane; ++i) {
      const float x = h_q32[i];
      const unsigned short hi = f2bf(x);
      h_qbf[i] = hi; h_qbf[plane + i] = f2bf(x - bf2f(hi));
    }
  }
  const char* pv = getenv("GOCTR_KNN_POLL_MAXQ");
  // (profiles/r05_knn_poll.txt: with the input over the BAR polling pays up to 64 queries per call -- the bench line 1.18 -> 1.26 M
  // queries/s --, is even at 96 and costs 3 us at 128; with the staged copy it lost 9 us at 64)
  // Without the BAR input (GOCTR_KNN_BAR=0, or no large BAR) the same polling lost 9 us at 64 queries: the default follows `bar`.
  const bool poll = Q <= (pv ? atoi(pv) : (bar ? 64 : 32));
  if (poll) {                                        // (the previous call returned after its kernels' last stores: nothing else writes here)
    int* h_pend = reinterpret_cast<int*>(static_cast<char*>(s->h_out) + o_idx + o_sim);
    for (int i = 0; i < Q; ++i) __atomic_store_n(h_pend + i, KNN_PENDING, __ATOMIC_RELEASE);
  }
  if (bar) {                                         // (the previous call's kernels have finished: the host waited for them)
    memcpy(d_in, s->h_in, in_bytes);
    __builtin_ia32_sfence();                         // write-combined stores out before the doorbell of the launch below
  } else GOCTR_HIP(hipMemcpyAsync(d_in, s->h_in, in_bytes, hipMemcpyHostToDevice, e.stream));
  const double* d_q = reinterpret_cast<const double*>(d_in);
  const long long* d_ig = reinterpret_cast<const long long*>(d_in + in_q);
  const float* d_q32 = reinterpret_cast<const float*>(d_in + in_q + in_ig);
  // the filter's error bound |a - sim|: float32 arithmetic (D + 8) 2^-23; bf16-plane arithmetic 2^-14 (knn_scan_bf16_kernel)
  const float E = bf ? 6.103515625e-05f : (float)(D + 8) * 1.1920929e-07f;
  char* d_out = nullptr;                              // the device's view of the pinned output buffer (zero-copy: 10 KB per call)
  GOCTR_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_out), s->h_out, 0));
  long long* d_oi = reinterpret_cast<long long*>(d_out);
  double* d_os = reinterpret_cast<double*>(d_out + o_idx);
  int* d_oc = reinterpret_cast<int*>(d_out + o_idx + o_sim);
#define GOCTR_KNN_SCAN(DD, IPT) hipLaunchKernelGGL((knn_scan_kernel<DD, IPT>), dim3(nt, nqb), dim3(256), 0, e.stream, s->items32.p, \
                                                  (long long)s->V, d_q32, Q, nt, s->tmax.p, s->bmax.p, qpad)
  if (bf) {
    const unsigned short* d_qbf = reinterpret_cast<const unsigned short*>(d_in + in_q + in_ig + in_q32);
    const long long qplane = (long long)nqb * KNN2_QB * D, iplane = (long long)(s->items_bf.n / 2);
    if (D == 16) hipLaunchKernelGGL(knn_scan_bf16_kernel<16>, dim3(nt, nqb), dim3(256), 0, e.stream, s->items_bf.p, iplane, d_qbf, qplane, Q, nt, s->tmax.p, s->bmax.p, qpad);
    else hipLaunchKernelGGL(knn_scan_bf16_kernel<32>, dim3(nt, nqb), dim3(256), 0, e.stream, s->items_bf.p, iplane, d_qbf, qplane, Q, nt, s->tmax.p, s->bmax.p, qpad);
  } else if (D == 16) GOCTR_KNN_SCAN(16, 4);
  else if (D == 32) GOCTR_KNN_SCAN(32, 4);
  else GOCTR_KNN_SCAN(64, 2);
#undef GOCTR_KNN_SCAN
  GOCTR_HIP(hipGetLastError());
  // collect + exact refine + replay: one workgroup per query (sub-block layout: 32 consecutive items from the matrix-core scan
  // kernels, the rows of 16 adjacent threads from the VALU one)
  const int sb_mode = bf ? 0 : 1;
  static DevBuf<unsigned long long> knn_dbg;
  const bool want_dbg = dbg_on("knn");
  if (want_dbg && !knn_dbg.p && knn_dbg.alloc(16)) return -1;
  unsigned long long* kdbg = want_dbg ? knn_dbg.p : nullptr;
  const size_t lds_q = sizeof(double) * ((size_t)D + 1 + ((size_t)D + 1) / 2);      // [D] query | norm | [D] floats (rounded up to doubles)
  hipLaunchKernelGGL(knn_collect_kernel, dim3(Q), dim3(256), lds_q + lds_r, e.stream, s->items.p, s->norms.p, s->items32.p, (long long)s->V, D,
                     d_q, d_q32, d_ig, s->tmax.p, s->bmax.p, qpad, nt, tile_items, sb_mode, k, E, d_oi, d_os, d_oc, poll ? 1 : 0, kdbg);
  GOCTR_HIP(hipGetLastError());
  if (kdbg) {
    unsigned long long h[10];
    if (knn_dbg.download(h, 10)) return -1;
    fprintf(stderr, "knn_collect query 0 (s_memtime ticks): query + tile maxima %llu | bound (rank of 256 group maxima) %llu | tile list %llu | "
            "sub-block maxima + list %llu | float32 filter + exact scores %llu | sort + replay %llu | results out %llu | total %llu "
            "(%llu candidates, %llu sub-blocks)\n", h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6],
            h[7] - h[0], h[8], h[9]);
  }
  const long long* h_oi = static_cast<const long long*>(s->h_out);
  const double* h_os = reinterpret_cast<const double*>(static_cast<const char*>(s->h_out) + o_idx);
  const int* h_oc = reinterpret_cast<const int*>(static_cast<const char*>(s->h_out) + o_idx + o_sim);
  // The collect kernel wrote the pinned host buffer itself (no copy command), each query's count last, behind a system-scope
  // release: the host watches the counts turn from the sentinel instead of waiting for the stream's completion signal.  A call
  // that has not finished after 2 ms (a long queue ahead of it, or a fault) falls back to the stream wait, which reports errors.
  bool polled = false;
  if (poll) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int done = 0, spins = 0; !polled;) {
      while (done < Q && __atomic_load_n(h_oc + done, __ATOMIC_ACQUIRE) != KNN_PENDING) ++done;
      if (done == Q) { polled = true; break; }
      if ((++spins & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
      __builtin_ia32_pause();
    }
  }
  if (!polled) GOCTR_HIP(hipStreamSynchronize(e.stream));
  for (int i = 0; i < Q; ++i) if (h_oc[i] < 0) return 1;
  for (size_t i = 0; i < (size_t)Q * k; ++i) { out_idx[i] = h_oi[i]; out_sim[i] = h_os[i]; }
  for (int i = 0; i < Q; ++i) out_count[i] = h_oc[i];
  return 0;
}

extern "C" {

int goctr_searcher_create(const double* items, int64_t V, int D, goctr_searcher** out) {
  GOCTR_ENTER();
  GOCTR_CHECK(items && out && V > 0 && D > 0, "goctr_searcher_create: bad arguments");
  GOCTR_CHECK(D <= 1024, "goctr_searcher_create: dim %d > 1024", D);
  GOCTR_CHECK(cdiv(V, KNN_TILE) <= 8192, "goctr_searcher_create: more than %d items", 8192 * KNN_TILE);
  goctr_searcher* s = new goctr_searcher;
  s->V = V; s->D = D;
  if (s->items.alloc((size_t)V * D, false) || s->items.upload(items, (size_t)V * D) || s->norms.alloc((size_t)V, false) ||
      (knn_scan_ipt(D) > 0 && s->items32.alloc((size_t)round_up64(V, 2048) * D, false)) ||
      ((D == 16 || D == 32) && s->items_bf.alloc((size_t)2 * round_up64(V, 2048) * D, true))) {      // (zeroed: the pad rows score 0)
    delete s;
    return -1;
  }
  if (s->items32.p)      // (zero rows up to a whole tile: the matrix-core scan kernel reads whole tiles)
    (void)hipMemsetAsync(s->items32.p + (size_t)V * D, 0, sizeof(float) * (size_t)(round_up64(V, 2048) - V) * D, engine().stream);
  hipLaunchKernelGGL(knn_norm_kernel, dim3((unsigned)cdiv(V, 256)), dim3(256), 0, engine().stream, s->items.p, (long long)V, D,
                     s->norms.p, s->items32.p, s->items_bf.p, (long long)(s->items_bf.n / 2));
  if (hipGetLastError() != hipSuccess) { set_error("knn_norm_kernel launch failed"); delete s; return -1; }
  *out = s;
  return 0;
}

void goctr_searcher_destroy(goctr_searcher* s) { delete s; }

// queries per device pass: the scratch of a pass grows with Q (sub-block maxima V x Q / 8 bytes, the tile kernels' candidate
// lists Q x tiles x k) -- V = 16 M with 4096 queries in one pass would need 8 GB that is never shrunk; a call with more
// queries is served in passes of this many (the queries are independent: same answers)
constexpr int KNN_PASS_MAXQ = 1024;
static int searcher_search_pass(goctr_searcher* s, const double* queries, int Q, int k, const int64_t* ignore, int64_t* out_idx,
                                double* out_sim, int* out_count);

int goctr_searcher_search(goctr_searcher* s, const double* queries, int Q, int k, const int64_t* ignore, int64_t* out_idx,
                          double* out_sim, int* out_count) {
  GOCTR_ENTER_H(s);
  GOCTR_CHECK(s && queries && out_idx && out_sim && out_count, "goctr_searcher_search: null argument");
  GOCTR_CHECK(Q > 0 && k > 0 && k <= KNN_MAX_K, "goctr_searcher_search: Q %d, k %d (k <= %d)", Q, k, KNN_MAX_K);
  std::lock_guard<std::mutex> lk(s->mu);
  for (int q0 = 0; q0 < Q; q0 += KNN_PASS_MAXQ) {
    const int n = Q - q0 < KNN_PASS_MAXQ ? Q - q0 : KNN_PASS_MAXQ;
    if (searcher_search_pass(s, queries + (size_t)q0 * s->D, n, k, ignore ? ignore + q0 : nullptr, out_idx + (size_t)q0 * k,
                             out_sim + (size_t)q0 * k, out_count + q0)) return -1;
  }
  return 0;
}

static int searcher_search_pass(goctr_searcher* s, const double* queries, int Q, int k, const int64_t* ignore, int64_t* out_idx,
                                double* out_sim, int* out_count) {
  Engine& e = engine();
  if (knn_scan_usable(s, k)) {
    const int rc = knn_search_scan(s, queries, Q, k, ignore, out_idx, out_sim, out_count);
    if (rc <= 0) return rc;                  // (1: a query had more candidates than the replay takes -- the tile kernels below)
  }
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
