/* TEST INFRASTRUCTURE ONLY - CPU oracle, never linked into or called by the product path.
 *
 * Restatement of the corpus load that precedes item2vec training, for integer tokens:
 *   memory.Corpus.Load          feature/embedding/corpus/memory/memory.go:76-102
 *   dictionary.Add / ID         feature/embedding/corpus/dictionary/dictionary.go:70-81, :43-46
 *   Corpus.IndexedDoc           memory.go:53-62
 *   cpsutil.MaxCount / MinCount feature/embedding/corpus/cpsutil/cpsutil.go:58-78
 *   subsample.New               feature/embedding/model/modelutil/subsample/subsample.go:28-43
 * Parity pin: the reference holds no golden vectors for these (cpsutil_test.go only checks ReadWord's order);
 * the KATs in tests/test_oracle_kats.py are derived by hand from the definitions above ("parity unpinned" beyond that).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "goctr_oracle.h"

/* Go's map[string]int, here int64 -> id; any correct map gives the same ids (they depend on arrival order only) */
typedef struct { int64_t key; int32_t id; } slot_t;

static uint64_t hash64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

/* Load: for every word  dic.Add(word); id, _ := dic.ID(word); maxLen++; idoc = append(idoc, id)
 * then IndexedDoc.  Returns V (dictionary size); idoc [n], id2key / cfs [<= n], indexed [<= n], *n_indexed. */
int64_t orc_corpus_build(const int64_t* keys, int64_t n, int64_t min_count, int64_t max_count, int32_t* idoc,
                         int64_t* id2key, int64_t* cfs, int32_t* indexed, int64_t* n_indexed) {
  uint64_t cap = 16;
  while (cap < (uint64_t)n * 2) cap <<= 1;
  slot_t* tab = (slot_t*)malloc(sizeof(slot_t) * cap);
  for (uint64_t i = 0; i < cap; i++) tab[i].id = -1;
  int64_t maxid = 0;
  for (int64_t pos = 0; pos < n; pos++) {
    const int64_t w = keys[pos];
    uint64_t h = hash64((uint64_t)w) & (cap - 1);
    while (tab[h].id >= 0 && tab[h].key != w) h = (h + 1) & (cap - 1);
    if (tab[h].id >= 0) {
      cfs[tab[h].id]++;                     /* dictionary.go:72-73 */
    } else {
      tab[h].key = w; tab[h].id = (int32_t)maxid;      /* :75 word2id[word] = maxid */
      id2key[maxid] = w;                    /* :76 */
      cfs[maxid] = 1;                       /* :77 */
      maxid++;                              /* :78 */
    }
    idoc[pos] = tab[h].id;                  /* memory.go:85-88 */
  }
  int64_t m = 0;
  for (int64_t pos = 0; pos < n; pos++) {   /* memory.go:53-62 with cpsutil.go:58-78 */
    const int64_t f = cfs[idoc[pos]];
    const int drop_max = 0 < max_count && max_count < f;
    const int drop_min = 0 <= min_count && f < min_count;
    if (drop_max || drop_min) continue;
    indexed[m++] = idoc[pos];
  }
  *n_indexed = m;
  free(tab);
  return maxid;
}

/* subsample.go:28-43 */
void orc_subsample_probs(const int64_t* cfs, int64_t V, double threshold, double* samples) {
  for (int64_t i = 0; i < V; i++) {
    double z = 1. - sqrt(threshold / (double)cfs[i]);
    if (z < 0) z = 0;
    samples[i] = z;
  }
}
