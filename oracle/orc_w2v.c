/*
 * orc_w2v.c -- oracle (TEST INFRASTRUCTURE, see goctr_oracle.h): float64 restatement of item2vec
 * (feature/embedding/..., vendored wego): LCG, sigmoid table, sub-sampling, Huffman tree + paths,
 * SkipGram / CBOW x hierarchical-softmax / negative-sampling updates, lr observer.
 *
 * The reference trains Hogwild over goroutines (racy by design, word2vec.go:151-175), seeds its
 * init / sub-sampling from Go's math/rand (not reproducible without Go) => "parity unpinned":
 * parity is defined on GIVEN init matrix, doc, keep-mask and the LCG stream, single stream.
 */
#define _GNU_SOURCE
#include "goctr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* modelutil.go:21-29 */
int orc_lcg_next(orc_lcg* g, int value) {
  g->next = g->next * (uint64_t)25214903917ULL + 11ULL;
  return (int)(g->next % (uint64_t)value);
}

/* modelutil.go:32-41 */
void orc_index_per_thread(int threads, int64_t n, int64_t* out) {
  out[0] = 0;
  out[threads] = n;
  for (int i = 1; i < threads; i++) out[i] = out[i - 1] + (int64_t)trunc((double)((n + i) / threads));
}

/* sigmoid_table.go:28-38 */
void orc_sigmoid_table(double* table) {
  const int size = 1000; const double max_exp = 6.0;
  for (int i = 0; i < size; i++) {
    double expval = exp(((double)i / (double)size * 2. - 1.) * max_exp);
    table[i] = expval / (expval + 1.);
  }
}
/* sigmoid_table.go:43-45 */
double orc_sigmoid_lookup(const double* table, double x) {
  const double cache = 1000.0 / 6.0 / 2.0;
  return table[(int)((x + 6.0) * cache)];
}

/* subsample.go:28-43 */
double orc_subsample_keep(double threshold, int64_t count) {
  double z = 1. - sqrt(threshold / (double)count);
  return z < 0 ? 0 : z;
}

/* ------------------------------------------------------------- huffman -- */
typedef struct { int64_t val; int32_t parent; uint8_t code; } hnode;

static int cmp_leaf(const void* a, const void* b, void* arg) {
  const int64_t* counts = (const int64_t*)arg;
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  if (counts[x] != counts[y]) return counts[x] < counts[y] ? -1 : 1;
  return x < y ? -1 : (x > y); /* stable: equal counts keep id order (sort.SliceStable) */
}

/* huffman.go:23-57 with the O(V^2) slice insertion replaced by a two-queue merge that keeps the
 * reference's tie-breaking: the merged node is inserted at the FIRST index whose Val >= merged.Val,
 * i.e. before every equal-valued node (leaf or earlier-merged).  Merged values are non-decreasing,
 * so merged nodes form a queue of equal-value runs, each run popped newest-first. */
int64_t orc_huffman_paths(const int64_t* counts, int64_t V, int max_depth,
                          int64_t* path_off, int32_t* path_nodes, uint8_t* path_codes, int64_t cap) {
  /* node ids: leaves 0..V-1, inner V..2V-2 (inner k = V+k, creation order) */
  int64_t total = 2 * V - 1;
  if (V <= 0) { path_off[0] = 0; return 0; }
  hnode* nd = (hnode*)calloc((size_t)total, sizeof(hnode));
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)V);
  for (int64_t i = 0; i < V; i++) { order[i] = (int32_t)i; nd[i].val = counts[i]; nd[i].parent = -1; }
  qsort_r(order, (size_t)V, sizeof(int32_t), cmp_leaf, (void*)counts);
  /* merged queue = runs of equal value (values are non-decreasing in creation order); each run
   * is a stack so the newest equal-valued node is consumed first */
  int64_t lq = 0; /* next leaf */
  /* runs stored as: run_val[], run_head[] (top of stack, index into mq order), linked by below[] */
  int64_t* run_val = (int64_t*)malloc(sizeof(int64_t) * (size_t)(V > 1 ? V : 1));
  int32_t* run_head = (int32_t*)malloc(sizeof(int32_t) * (size_t)(V > 1 ? V : 1));
  int32_t* below = (int32_t*)malloc(sizeof(int32_t) * (size_t)(V > 1 ? V : 1));
  int64_t rfront = 0, rback = 0; /* runs [rfront, rback) */
  for (int64_t k = 0; k < V - 1; k++) {
    int32_t pick[2];
    for (int s = 0; s < 2; s++) {
      int have_leaf = lq < V, have_m = rfront < rback;
      int take_m;
      if (have_leaf && have_m) take_m = run_val[rfront] <= nd[order[lq]].val; /* merged sits before equal leaves */
      else take_m = have_m;
      if (take_m) {
        int32_t idx = run_head[rfront];       /* index k' of inner node */
        pick[s] = (int32_t)(V + idx);
        run_head[rfront] = below[idx];
        if (run_head[rfront] < 0) rfront++;
      } else {
        pick[s] = order[lq++];
      }
    }
    int32_t id = (int32_t)(V + k);
    nd[id].val = nd[pick[0]].val + nd[pick[1]].val;
    nd[id].parent = -1;
    nd[pick[0]].code = 0; nd[pick[1]].code = 1;
    nd[pick[0]].parent = id; nd[pick[1]].parent = id;
    if (rfront < rback && run_val[rback - 1] == nd[id].val) { /* joins the newest run, on top */
      below[k] = run_head[rback - 1];
      run_head[rback - 1] = (int32_t)k;
    } else {
      run_val[rback] = nd[id].val; run_head[rback] = (int32_t)k; below[k] = -1; rback++;
    }
  }
  /* paths: node.go:26-43 */
  int64_t w = 0;
  int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)(total + 1));
  for (int64_t i = 0; i < V; i++) {
    path_off[i] = w;
    int64_t len = 0;
    for (int32_t p = (int32_t)i; p >= 0; p = nd[p].parent) tmp[len++] = p; /* leaf..root */
    int64_t depth = max_depth < len ? max_depth : len;                     /* cache[:depth], root first */
    for (int64_t j = 0; j + 1 < depth; j++) {
      int32_t pn = tmp[len - 1 - j], child = tmp[len - 2 - j];
      if (w < cap) { path_nodes[w] = pn - (int32_t)V; path_codes[w] = nd[child].code; }
      w++;
    }
  }
  path_off[V] = w;
  free(nd); free(order); free(run_val); free(run_head); free(below); free(tmp);
  return w;
}

/* Literal O(V^2) restatement of huffman.go:23-57 (sorted slice + sort.Search insertion); used by
 * the tests to pin the fast builder above on inputs with many ties. */
int64_t orc_huffman_paths_slow(const int64_t* counts, int64_t V, int max_depth,
                               int64_t* path_off, int32_t* path_nodes, uint8_t* path_codes, int64_t cap) {
  int64_t total = 2 * V - 1;
  if (V <= 0) { path_off[0] = 0; return 0; }
  hnode* nd = (hnode*)calloc((size_t)total, sizeof(hnode));
  int32_t* list = (int32_t*)malloc(sizeof(int32_t) * (size_t)(V + 1));
  for (int64_t i = 0; i < V; i++) { list[i] = (int32_t)i; nd[i].val = counts[i]; nd[i].parent = -1; }
  qsort_r(list, (size_t)V, sizeof(int32_t), cmp_leaf, (void*)counts);
  int64_t n = V, created = 0;
  while (n > 1) {
    int32_t left = list[0], right = list[1];
    int32_t id = (int32_t)(V + created++);
    nd[id].val = nd[left].val + nd[right].val; nd[id].parent = -1;
    nd[left].code = 0; nd[right].code = 1;
    nd[left].parent = id; nd[right].parent = id;
    memmove(list, list + 2, sizeof(int32_t) * (size_t)(n - 2));
    n -= 2;
    int64_t lo = 0, hi = n; /* sort.Search: first i with list[i].val >= merged.val */
    while (lo < hi) { int64_t mid = lo + (hi - lo) / 2; if (nd[list[mid]].val >= nd[id].val) hi = mid; else lo = mid + 1; }
    memmove(list + lo + 1, list + lo, sizeof(int32_t) * (size_t)(n - lo));
    list[lo] = id; n++;
  }
  int64_t w = 0;
  int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)(total + 1));
  for (int64_t i = 0; i < V; i++) {
    path_off[i] = w;
    int64_t len = 0;
    for (int32_t p = (int32_t)i; p >= 0; p = nd[p].parent) tmp[len++] = p;
    int64_t depth = max_depth < len ? max_depth : len;
    for (int64_t j = 0; j + 1 < depth; j++) {
      int32_t pn = tmp[len - 1 - j], child = tmp[len - 2 - j];
      if (w < cap) { path_nodes[w] = pn - (int32_t)V; path_codes[w] = nd[child].code; }
      w++;
    }
  }
  path_off[V] = w;
  free(nd); free(list); free(tmp);
  return w;
}

/* ------------------------------------------------------------ training -- */
typedef struct {
  const orc_w2v_cfg* cfg;
  double* param; double* aux; int64_t V;
  const int64_t* path_off; const int32_t* path_nodes; const uint8_t* path_codes;
  const double* sigtab; orc_lcg* lcg;
} w2v_ctx;

/* optimizer.go:107-129 */
static void hs_optim(const w2v_ctx* c, int id, double lr, const double* ctx, double* tmp) {
  const int dim = c->cfg->dim;
  for (int64_t i = c->path_off[id]; i < c->path_off[id + 1]; i++) {
    double* pv = c->aux + (int64_t)c->path_nodes[i] * dim;
    const int code = c->path_codes[i];
    double inner = 0;
    for (int j = 0; j < dim; j++) inner += ctx[j] * pv[j];
    if (inner <= -6.0 || inner >= 6.0) return; /* Q13: return, not continue */
    const double g = (1.0 - (double)code - orc_sigmoid_lookup(c->sigtab, inner)) * lr;
    for (int j = 0; j < dim; j++) {
      tmp[j] += g * pv[j];
      pv[j] += g * ctx[j];
    }
  }
}

/* optimizer.go:52-91 */
static void ns_optim(const w2v_ctx* c, int id, double lr, const double* ctx, double* tmp) {
  const int dim = c->cfg->dim;
  for (int n = -1; n < c->cfg->neg_samples; n++) {
    int label, picked;
    if (n == -1) { label = 1; picked = id; }
    else {
      label = 0;
      picked = orc_lcg_next(c->lcg, (int)c->V);
      if (id == picked) continue;
    }
    double* rnd = c->aux + (int64_t)picked * dim;
    double inner = 0;
    for (int i = 0; i < dim; i++) inner += rnd[i] * ctx[i];
    double g;
    if (inner <= -6.0) g = ((double)(label - 0)) * lr;
    else if (inner >= 6.0) g = ((double)(label - 1)) * lr;
    else g = ((double)label - orc_sigmoid_lookup(c->sigtab, inner)) * lr;
    for (int i = 0; i < dim; i++) {
      tmp[i] += g * rnd[i];
      rnd[i] += g * ctx[i];
    }
  }
}

static inline void optim(const w2v_ctx* c, int id, double lr, const double* ctx, double* tmp) {
  if (c->cfg->optimizer == 0) hs_optim(c, id, lr, ctx, tmp); else ns_optim(c, id, lr, ctx, tmp);
}

/* model.go:48-78 */
static void skipgram_one(const w2v_ctx* c, const int32_t* doc, int64_t len, int64_t pos, double lr, double* tmp) {
  const int win = c->cfg->window, dim = c->cfg->dim;
  int del = orc_lcg_next(c->lcg, win);
  for (int a = del; a < win * 2 + 1 - del; a++) {
    if (a == win) continue;
    int64_t cpos = pos - win + a;
    if (cpos < 0 || cpos >= len) continue;
    for (int i = 0; i < dim; i++) tmp[i] = 0;
    double* ctx = c->param + (int64_t)doc[cpos] * dim;
    optim(c, doc[pos], lr, ctx, tmp);
    for (int i = 0; i < dim; i++) ctx[i] += tmp[i];
  }
}

/* model.go:96-148 (note: dowith draws a fresh NextRandom for aggregate AND for update) */
static void cbow_one(const w2v_ctx* c, const int32_t* doc, int64_t len, int64_t pos, double lr, double* agg, double* tmp) {
  const int win = c->cfg->window, dim = c->cfg->dim;
  for (int i = 0; i < dim; i++) { agg[i] = 0; tmp[i] = 0; }
  int del = orc_lcg_next(c->lcg, win);
  for (int a = del; a < win * 2 + 1 - del; a++) {
    if (a == win) continue;
    int64_t cpos = pos - win + a;
    if (cpos < 0 || cpos >= len) continue;
    const double* ctx = c->param + (int64_t)doc[cpos] * dim;
    for (int i = 0; i < dim; i++) agg[i] += ctx[i];
  }
  optim(c, doc[pos], lr, agg, tmp);
  del = orc_lcg_next(c->lcg, win);
  for (int a = del; a < win * 2 + 1 - del; a++) {
    if (a == win) continue;
    int64_t cpos = pos - win + a;
    if (cpos < 0 || cpos >= len) continue;
    double* ctx = c->param + (int64_t)doc[cpos] * dim;
    for (int i = 0; i < dim; i++) ctx[i] += tmp[i];
  }
}

/* word2vec.go:198-243 for one goroutine's slice, with the observer folded in */
void orc_w2v_train_slice(const orc_w2v_cfg* cfg, const int32_t* doc, int64_t lo, int64_t hi,
                         const uint8_t* keep_mask, double* param, double* aux, int64_t V,
                         const int64_t* path_off, const int32_t* path_nodes, const uint8_t* path_codes,
                         const double* sigtab, orc_lcg* lcg,
                         double* lr, int64_t* trained_cnt, int64_t corpus_len) {
  w2v_ctx c = {cfg, param, aux, V, path_off, path_nodes, path_codes, sigtab, lcg};
  double tmp[1024], agg[1024];
  const int32_t* sl = doc + lo;
  const int64_t len = hi - lo;
  for (int64_t pos = 0; pos < len; pos++) {
    if (!keep_mask || keep_mask[lo + pos]) {
      if (cfg->model == 0) skipgram_one(&c, sl, len, pos, *lr, tmp);
      else cbow_one(&c, sl, len, pos, *lr, agg, tmp);
    }
    /* observe(): word2vec.go:223-233 */
    int64_t cnt = ++(*trained_cnt);
    if (cnt % cfg->update_lr_batch == 0) {
      if (*lr < cfg->min_lr) *lr = cfg->min_lr;
      else *lr = cfg->init_lr * (1.0 - (double)cnt / (double)corpus_len);
    }
  }
}

/* The same walk over positions [walk_lo, walk_hi) of a slice [clip_lo, clip_hi): windows are clipped at the SLICE's ends
 * (model.go:60-77 clips to the goroutine's slice), not at the walked range's -- what one SEGMENT of a data-parallel pass of the
 * device library does (csrc/w2v.hip: a pass is cut into segments with a parameter exchange after each; a stream's windows still
 * reach into the neighbouring segments of its own piece).  Test infrastructure for tests/test_gpu_multi.py. */
void orc_w2v_train_range(const orc_w2v_cfg* cfg, const int32_t* doc, int64_t clip_lo, int64_t clip_hi, int64_t walk_lo,
                         int64_t walk_hi, const uint8_t* keep_mask, double* param, double* aux, int64_t V,
                         const int64_t* path_off, const int32_t* path_nodes, const uint8_t* path_codes,
                         const double* sigtab, orc_lcg* lcg, double* lr, int64_t* trained_cnt, int64_t corpus_len) {
  w2v_ctx c = {cfg, param, aux, V, path_off, path_nodes, path_codes, sigtab, lcg};
  double tmp[1024], agg[1024];
  const int32_t* sl = doc + clip_lo;
  const int64_t len = clip_hi - clip_lo;
  for (int64_t pos = walk_lo - clip_lo; pos < walk_hi - clip_lo; pos++) {
    if (!keep_mask || keep_mask[clip_lo + pos]) {
      if (cfg->model == 0) skipgram_one(&c, sl, len, pos, *lr, tmp);
      else cbow_one(&c, sl, len, pos, *lr, agg, tmp);
    }
    int64_t cnt = ++(*trained_cnt);
    if (cnt % cfg->update_lr_batch == 0) {
      if (*lr < cfg->min_lr) *lr = cfg->min_lr;
      else *lr = cfg->init_lr * (1.0 - (double)cnt / (double)corpus_len);
    }
  }
}

/* Hogwild CPU baseline: word2vec.go:151-175 (threads = goroutines). Shared, unsynchronised
 * param / aux / LCG / lr, as in the reference; the per-word channel send is replaced by an atomic
 * counter (a faster observer than the reference's unbuffered channel). */
void orc_w2v_train_hogwild(const orc_w2v_cfg* cfg, const int32_t* doc, int64_t n, int threads,
                           const uint8_t* keep_mask, double* param, double* aux, int64_t V,
                           const int64_t* path_off, const int32_t* path_nodes, const uint8_t* path_codes,
                           const double* sigtab, double* lr, int64_t corpus_len) {
  int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)(threads + 1));
  orc_index_per_thread(threads, n, idx);
  orc_lcg lcg = {1};
  int64_t cnt_shared = 0;
  volatile double* lrp = lr;
#pragma omp parallel num_threads(threads)
  {
#ifdef _OPENMP
    int t = omp_get_thread_num();
#else
    int t = 0;
#endif
    w2v_ctx c = {cfg, param, aux, V, path_off, path_nodes, path_codes, sigtab, &lcg};
    double tmp[1024], agg[1024];
    const int32_t* sl = doc + idx[t];
    const int64_t len = idx[t + 1] - idx[t];
    for (int64_t pos = 0; pos < len; pos++) {
      if (!keep_mask || keep_mask[idx[t] + pos]) {
        if (cfg->model == 0) skipgram_one(&c, sl, len, pos, *lrp, tmp);
        else cbow_one(&c, sl, len, pos, *lrp, agg, tmp);
      }
      int64_t cnt = __atomic_add_fetch(&cnt_shared, 1, __ATOMIC_RELAXED);
      if (cnt % cfg->update_lr_batch == 0) {
        if (*lrp < cfg->min_lr) *lrp = cfg->min_lr;
        else *lrp = cfg->init_lr * (1.0 - (double)cnt / (double)corpus_len);
      }
    }
  }
  free(idx);
}
