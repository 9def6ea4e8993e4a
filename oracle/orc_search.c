/* orc_search.c -- CPU restatement of go-ctr's embedding k-NN search (TEST INFRASTRUCTURE ONLY, see
 * goctr_oracle.h): feature/embedding/search/search.go:92-134 (Searcher.Search),
 * search/searchutil/searchutil.go:17-26 (Cosine), emb/embutil/embutil.go:21-27 (Norm).
 * Pinned by the reference's own KATs (searchutil_test.go TestCosine, search_test.go TestSearchInternal /
 * TestSearchVector), transcribed as data in tests/golden/ref_kats.json. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "goctr_oracle.h"

/* embutil.go:21-27 */
double orc_norm64(const double* v, int d) {
  double n = 0;
  for (int i = 0; i < d; ++i) n += v[i] * v[i];
  return sqrt(n);
}

/* searchutil.go:17-26 */
double orc_cosine64(const double* v1, const double* v2, int d, double n1, double n2) {
  if (n1 == 0 || n2 == 0) return 0;
  double dot = 0;
  for (int i = 0; i < d; ++i) dot += v1[i] * v2[i];
  return dot / n1 / n2;
}

/* search.go:92-134.  items [V,D] row-major, norms [V]; ignore = index of the item to skip (SearchInternal passes the
 * query word) or -1.  out_idx/out_sim/out_rank have k entries; empty neighbours are idx -1, sim 0, rank 0 (the Go
 * zero value).  Returns the length of the returned slice, INCLUDING the reference's tail quirk: the guard loop
 * `if neighbors[i].Word == "" { k = i }` keeps overwriting k, so with e < k filled entries the slice has k-1
 * entries (the last k-1-e of them empty), not e. */
int orc_knn_search(const double* items, const double* norms, int64_t V, int D, const double* query, double qnorm,
                   int k, int64_t ignore, int64_t* out_idx, double* out_sim, int* out_rank) {
  for (int i = 0; i < k; ++i) { out_idx[i] = -1; out_sim[i] = 0; out_rank[i] = 0; }
  if (k <= 0) return 0;
  double low = .0;
  for (int64_t it = 0; it < V; ++it) {
    if (it == ignore) continue;
    const double score = orc_cosine64(query, items + (size_t)it * D, D, qnorm, norms[it]);
    if (score > low) {
      int64_t tidx = it; double tsim = score; int trank = 0;
      for (int i = 0; i < k; ++i) {
        if (tsim > out_sim[i]) {
          const int64_t xi = out_idx[i]; const double xs = out_sim[i]; const int xr = out_rank[i];
          out_idx[i] = tidx; out_sim[i] = tsim; out_rank[i] = trank;
          tidx = xi; tsim = xs; trank = xr;
          out_rank[i] = i + 1;
        }
      }
      low = out_sim[k - 1];
    }
  }
  int kk = k;
  for (int i = 0; i < k; ++i)
    if (out_idx[i] < 0) kk = i;
  return kk;
}

/* Q independent searches (the reference serves them from concurrent gin handler goroutines, recommend/api.go:106-131): OpenMP over
 * the QUERIES, each the sequential loop above -- no arithmetic changes.  out_* are [Q][k]; out_count [Q]. */
void orc_knn_search_batch(const double* items, const double* norms, int64_t V, int D, const double* queries, int Q, int k,
                          const int64_t* ignore, int64_t* out_idx, double* out_sim, int* out_rank, int* out_count) {
#pragma omp parallel for num_threads(orc_get_threads()) schedule(dynamic, 1)
  for (int q = 0; q < Q; ++q) {
    const double* qv = queries + (size_t)q * D;
    out_count[q] = orc_knn_search(items, norms, V, D, qv, orc_norm64(qv, D), k, ignore ? ignore[q] : -1, out_idx + (size_t)q * k,
                                  out_sim + (size_t)q * k, out_rank + (size_t)q * k);
  }
}
