/* orc_ubcache.c -- CPU restatement of go-ctr's user-behaviour cache lookup and of the per-sample row assembly that
 * consumes it (TEST INFRASTRUCTURE ONLY, see goctr_oracle.h):
 *   feature/ubcache/cache.go:71-94   TimeSeq.Filter (sequence in timestamp-descending order)
 *   recommend/rcmd.go:460-536        GetSampleVector: [userFeature | behaviour item ids -> embeddings | itemEmb | itemFeature]
 * Pinned by the reference's own KATs (feature/ubcache/cache_test.go), transcribed in tests/golden/ref_kats.json. */
#include <stdint.h>

#include "goctr_oracle.h"

/* cache.go:71-94.  ts/items: one user's sequence, newest first.  Writes up to max_len item ids to out (all of the
 * remaining ones when max_len == 0) and returns how many; out may be NULL to just count. */
int64_t orc_ubcache_filter(const int64_t* ts, const int32_t* items, int64_t len, int64_t max_ts, int64_t max_len,
                           int32_t* out) {
  if (len <= 0) return 0;                  /* (the Go code would index Ts[0] of an empty sequence; callers never store one) */
  if (max_ts == 0) max_ts = ts[0];
  int64_t count = max_len;
  if (count == 0) count = len;
  int64_t i;
  for (i = 0; i < len; ++i)
    if (ts[i] <= max_ts) break;
  if (i + count > len) count = len - i;
  if (out)
    for (int64_t j = 0; j < count; ++j) out[j] = items[i + j];
  return count;
}

/* rcmd.go:460-536 in id form: per sample (user, item, ts) the behaviour slots = Filter(ts, T) padded with -1 (the
 * reference leaves the embedding slots of missing behaviours zero), user / item feature rows copied from the tables. */
void orc_assemble_keys(const int64_t* off, const int32_t* seq_items, const int64_t* seq_ts, int64_t n_users,
                       const float* user_table, int U, const float* item_table, int64_t n_items, int C,
                       const int32_t* users, const int32_t* items, const int64_t* ts, int64_t rows, int T,
                       int32_t* ub_ids, float* ufeat, float* cfeat) {
  for (int64_t r = 0; r < rows; ++r) {
    int32_t* o = ub_ids + r * T;
    for (int j = 0; j < T; ++j) o[j] = -1;
    const int32_t u = users[r];
    if (u >= 0 && u < n_users)
      orc_ubcache_filter(seq_ts + off[u], seq_items + off[u], off[u + 1] - off[u], ts[r], T, o);
    for (int j = 0; j < U; ++j) ufeat[r * U + j] = (u >= 0 && u < n_users) ? user_table[(int64_t)u * U + j] : 0.f;
    const int32_t it = items[r];
    for (int j = 0; j < C; ++j) cfeat[r * C + j] = (it >= 0 && it < n_items) ? item_table[(int64_t)it * C + j] : 0.f;
  }
}
