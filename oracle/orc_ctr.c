/*
 * orc_ctr.c -- oracle (TEST INFRASTRUCTURE, see goctr_oracle.h): float32 restatement of the
 * DIN / YouTube-DNN forward, the hand-derived backward, gorgonia's AdamSolver step and the
 * model.Train / model.Predict mini-batch loops.
 *
 * Reference: model/din/din.go:219-323, model/youtube/dnn.go:162-184, model/activation.go:57-83,
 * model/cost.go:9-17, model/model.go:27-213,242-371, recommend/rcmd.go:462-536.
 * Third-party semantics (gorgonia v0.9.17 sigmoid clamp, Dropout, AdamSolver; go.mod:26-27) are
 * restated FROM MEMORY -- "parity unpinned" for the full step.
 *
 * Row loops are OpenMP-parallel in a way that never changes a summation order: per-row work is
 * independent, weight-gradient reductions are parallel over OUTPUT elements and sequential over
 * rows.  Results are identical for any thread count.
 */
#include "goctr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int orc_get_threads(void) { return g_threads; }

/* gorgonia _sigmoidf32 [from memory]: clamps, f64 exp, rounded to f32 */
static inline float sigm32(float x) {
  if (x < -88.f) return 0.f;
  if (x > 15.f) return 1.f;
  return (float)(1.0 / (1.0 + exp((double)(-x))));
}

static inline uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
/* counter-hash dropout mask shared bit-for-bit with goctr_amd/csrc (dropout mode 2).
 * gorgonia Dropout [from memory]: mask = (U(0,1) < keep), y = x*mask/keep. */
float orc_dropout_keep(uint32_t seed, uint32_t step, uint32_t layer, uint32_t row, uint32_t col, float p) {
  uint32_t h = mix32(seed ^ 0x9E3779B9u);
  h = mix32(h ^ (step * 2u + layer));
  h = mix32(h ^ row);
  h = mix32(h ^ (col * 0x85EBCA6Bu + 0xC2B2AE35u));
  float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  return u < (1.0f - p) ? 1.0f : 0.0f;
}

/* recommend/rcmd.go:462-536 + utils/util.go:22-28 */
void orc_assemble_rows(const float* emb, int64_t V, int D, int T,
                       const int32_t* ub_ids, const int32_t* item_ids,
                       const float* user_feat, int U, const float* item_feat, int C,
                       int64_t rows, float* X) {
  int xcols = U + T * D + D + C;
  for (int64_t r = 0; r < rows; r++) {
    float* row = X + r * xcols;
    memcpy(row, user_feat + r * U, sizeof(float) * (size_t)U);
    float* ub = row + U;
    memset(ub, 0, sizeof(float) * (size_t)(T * D + D));
    if (emb) {
      for (int t = 0; t < T; t++) {
        int32_t id = ub_ids[r * T + t];
        if (id >= 0 && id < V) memcpy(ub + t * D, emb + (int64_t)id * D, sizeof(float) * (size_t)D);
      }
      int32_t it = item_ids[r];
      if (it >= 0 && it < V) memcpy(ub + T * D, emb + (int64_t)it * D, sizeof(float) * (size_t)D);
    }
    memcpy(row + U + T * D + D, item_feat + r * C, sizeof(float) * (size_t)C);
  }
}

typedef struct {
  int B, I;
  float *h0, *A0, *A1, *K0, *K1, *P0, *P1, *y, *gate, *wgt;
} fwd_bufs;

/* forward of ONE row into row-b slots of the buffers. xr==NULL => all-zero (padding) row. */
static void fwd_row(const orc_ctr_cfg* cfg, const orc_ctr_weights* w, const float* xr, const int ranges[8],
                    int b, const orc_dropout* drop, fwd_bufs* fb) {
  const int U = cfg->U, T = cfg->T, D = cfg->D, C = cfg->C, H1 = cfg->H1, H2 = cfg->H2;
  const int I = U + 2 * D + C;
  float* h0 = fb->h0 + (size_t)b * I;
  float* gate = fb->gate + (size_t)b * T;
  float* wgt = fb->wgt + (size_t)b * T;
  if (!xr) {
    memset(h0, 0, sizeof(float) * (size_t)I);
  } else {
    memcpy(h0, xr + ranges[0], sizeof(float) * (size_t)U);
    memcpy(h0 + U + D, xr + ranges[4], sizeof(float) * (size_t)D);
    memcpy(h0 + U + 2 * D, xr + ranges[6], sizeof(float) * (size_t)C);
  }
  const float* v = h0 + U + D;
  float* p = h0 + U;
  for (int d = 0; d < D; d++) p[d] = 0.f;
  float syy = 0.f;
  for (int d = 0; d < D; d++) syy += v[d] * v[d];
  const float yn = sqrtf(syy);
  for (int t = 0; t < T; t++) {
    float xt[256];
    for (int d = 0; d < D; d++) xt[d] = xr ? xr[ranges[2] + t * D + d] : 0.f;
    float g;
    if (cfg->kind == ORC_DIN) {
      float wv;
      if (cfg->att == ORC_ATT_COSINE) { /* activation.go:57-83, din.go:231-237 */
        float sxx = 0.f, sxy = 0.f;
        for (int d = 0; d < D; d++) { sxx += xt[d] * xt[d]; sxy += xt[d] * v[d]; }
        float cosv = sxy / (sqrtf(sxx) * yn + 1e-8f);
        wv = (cosv + 1.0f) / 2.0f;
      } else { /* din.go:230 (commented-out variant): 1 - EucDistance */
        float s = 0.f;
        for (int d = 0; d < D; d++) { float df = xt[d] - v[d]; s += df * df; }
        wv = 1.0f - sqrtf(s);
      }
      g = sigm32(wv * w->att0[t]); /* din.go:264-276 */
      wgt[t] = wv;
    } else { /* dnn.go:167 plain mean */
      g = 1.0f; wgt[t] = 0.f;
    }
    gate[t] = g;
    for (int d = 0; d < D; d++) p[d] += g * xt[d]; /* Sum over axis 1 ... */
  }
  for (int d = 0; d < D; d++) p[d] = p[d] / (float)T; /* ... / T  (G.Mean, din.go:298) */

  float* A0 = fb->A0 + (size_t)b * H1;
  float* K0 = fb->K0 + (size_t)b * H1;
  float z0[1024];
  /* i-outer / j-inner: per-j summation order is still i = 0..I-1 (same values as the naive
   * dot product), but the inner loop is unit-stride so the CPU baseline is not crippled */
  for (int j = 0; j < H1; j++) z0[j] = 0.f;
  for (int i = 0; i < I; i++) {
    const float a = h0[i];
    const float* wr = w->W0 + (size_t)i * H1;
    for (int j = 0; j < H1; j++) z0[j] += a * wr[j];
  }
  for (int j = 0; j < H1; j++) {
    float s = z0[j];
    float a = sigm32(s); /* din.go:307 */
    fb->P0[(size_t)b * H1 + j] = a;
    float k = 1.0f;      /* dropout scale factor mask/keep (din.go:308) */
    if (drop && drop->mode && drop->p0 > 0.f) {
      float keep = 1.0f - drop->p0;
      float m = drop->mode == 1 ? drop->m0[(size_t)b * H1 + j]
                                : orc_dropout_keep(drop->seed, drop->step, 0, (uint32_t)b, (uint32_t)j, drop->p0);
      a = a * m / keep;
      k = m / keep;
    }
    A0[j] = a; K0[j] = k;
  }
  float* A1 = fb->A1 + (size_t)b * H2;
  float* K1 = fb->K1 + (size_t)b * H2;
  float z1[1024];
  for (int j = 0; j < H2; j++) z1[j] = 0.f;
  for (int i = 0; i < H1; i++) {
    const float a = A0[i];
    const float* wr = w->W1 + (size_t)i * H2;
    for (int j = 0; j < H2; j++) z1[j] += a * wr[j];
  }
  for (int j = 0; j < H2; j++) {
    float s = z1[j];
    float a = sigm32(s); /* din.go:311 */
    fb->P1[(size_t)b * H2 + j] = a;
    float k = 1.0f;
    if (drop && drop->mode && drop->p1 > 0.f) {
      float keep = 1.0f - drop->p1;
      float m = drop->mode == 1 ? drop->m1[(size_t)b * H2 + j]
                                : orc_dropout_keep(drop->seed, drop->step, 1, (uint32_t)b, (uint32_t)j, drop->p1);
      a = a * m / keep;
      k = m / keep;
    }
    A1[j] = a; K1[j] = k;
  }
  float s = 0.f;
  for (int i = 0; i < H2; i++) s += A1[i] * w->W2[i];
  fb->y[b] = sigm32(s); /* din.go:315 */
}

static fwd_bufs alloc_bufs(const orc_ctr_cfg* cfg, int B) {
  fwd_bufs fb;
  fb.B = B; fb.I = cfg->U + 2 * cfg->D + cfg->C;
  fb.h0 = (float*)calloc((size_t)B * fb.I, sizeof(float));
  fb.A0 = (float*)calloc((size_t)B * cfg->H1, sizeof(float));
  fb.K0 = (float*)calloc((size_t)B * cfg->H1, sizeof(float));
  fb.A1 = (float*)calloc((size_t)B * cfg->H2, sizeof(float));
  fb.K1 = (float*)calloc((size_t)B * cfg->H2, sizeof(float));
  fb.P0 = (float*)calloc((size_t)B * cfg->H1, sizeof(float));
  fb.P1 = (float*)calloc((size_t)B * cfg->H2, sizeof(float));
  fb.y = (float*)calloc((size_t)B, sizeof(float));
  fb.gate = (float*)calloc((size_t)B * cfg->T, sizeof(float));
  fb.wgt = (float*)calloc((size_t)B * cfg->T, sizeof(float));
  return fb;
}
static void free_bufs(fwd_bufs* fb) {
  free(fb->h0); free(fb->A0); free(fb->K0); free(fb->A1); free(fb->K1); free(fb->P0); free(fb->P1); free(fb->y); free(fb->gate); free(fb->wgt);
}

static void forward_batch(const orc_ctr_cfg* cfg, const orc_ctr_weights* w, const float* X, int xcols,
                          const int ranges[8], int B, int valid, const orc_dropout* drop, fwd_bufs* fb) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int b = 0; b < B; b++)
    fwd_row(cfg, w, b < valid ? X + (size_t)b * xcols : NULL, ranges, b, drop, fb);
}

void orc_ctr_forward(const orc_ctr_cfg* cfg, const orc_ctr_weights* w,
                     const float* X, int xcols, const int ranges[8], int B, int valid,
                     const orc_dropout* drop,
                     float* y_out, float* h0, float* A0, float* A1, float* gate, float* wgt) {
  fwd_bufs fb = alloc_bufs(cfg, B);
  forward_batch(cfg, w, X, xcols, ranges, B, valid, drop, &fb);
  memcpy(y_out, fb.y, sizeof(float) * (size_t)B);
  if (h0) memcpy(h0, fb.h0, sizeof(float) * (size_t)B * fb.I);
  if (A0) memcpy(A0, fb.A0, sizeof(float) * (size_t)B * cfg->H1);
  if (A1) memcpy(A1, fb.A1, sizeof(float) * (size_t)B * cfg->H2);
  if (gate) memcpy(gate, fb.gate, sizeof(float) * (size_t)B * cfg->T);
  if (wgt) memcpy(wgt, fb.wgt, sizeof(float) * (size_t)B * cfg->T);
  free_bufs(&fb);
}

/* forward + BCE + backward for one batch (SURVEY App. A.1 "Backward") */
float orc_ctr_loss_grad(const orc_ctr_cfg* cfg, const orc_ctr_weights* w,
                        const float* X, int xcols, const int ranges[8],
                        const float* Y, int B, int valid, const orc_dropout* drop,
                        orc_ctr_weights* grads, float* y_out) {
  const int U = cfg->U, T = cfg->T, D = cfg->D, C = cfg->C, H1 = cfg->H1, H2 = cfg->H2;
  const int I = U + 2 * D + C;
  fwd_bufs fb = alloc_bufs(cfg, B);
  forward_batch(cfg, w, X, xcols, ranges, B, valid, drop, &fb);
  if (y_out) memcpy(y_out, fb.y, sizeof(float) * (size_t)B);

  float* yv = (float*)calloc((size_t)B, sizeof(float));
  for (int b = 0; b < valid; b++) yv[b] = Y[b];
  const float cost = orc_bce32(fb.y, yv, B); /* cost.go:9-17 over the PADDED batch (Q3) */

  float* dz2 = (float*)malloc(sizeof(float) * (size_t)B);
  float* dz1 = (float*)malloc(sizeof(float) * (size_t)B * H2);
  float* dz0 = (float*)malloc(sizeof(float) * (size_t)B * H1);
  float* dgs = (float*)calloc((size_t)B * T, sizeof(float)); /* dg*g(1-g)*w per (b,t) */
  const float one_eps = (float)(1.0 + 1e-8);
  const float invB = 1.0f / (float)B;

#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int b = 0; b < B; b++) {
    const float p = fb.y[b], y = yv[b];
    /* d cost / d p = -(1/B) * ( y/p - (1-y)/(c-p) ) */
    float dy = -((y / p) - ((1.0f - y) / (one_eps - p))) * invB;
    float d2 = dy * (p * (1.0f - p));
    dz2[b] = d2;
    const float* K1 = fb.K1 + (size_t)b * H2;
    float* r1 = dz1 + (size_t)b * H2;
    for (int j = 0; j < H2; j++) {
      float dA1t = d2 * w->W2[j];     /* d wrt post-dropout activation */
      float k = K1[j];
      float a_pre = fb.P1[(size_t)b * H2 + j]; /* pre-dropout sigmoid output */
      float dA1 = dA1t * k;           /* A1 = a_pre*m/keep => grad wrt a_pre = dA1t*m/keep */
      r1[j] = dA1 * (a_pre * (1.0f - a_pre));
    }
    const float* K0 = fb.K0 + (size_t)b * H1;
    float* r0 = dz0 + (size_t)b * H1;
    for (int i = 0; i < H1; i++) {
      float s = 0.f;
      for (int j = 0; j < H2; j++) s += r1[j] * w->W1[(size_t)i * H2 + j];
      float k = K0[i];
      float a_pre = fb.P0[(size_t)b * H1 + i];
      float dA0 = s * k;
      r0[i] = dA0 * (a_pre * (1.0f - a_pre));
    }
    if (cfg->kind == ORC_DIN) {
      float dpT[256];
      for (int d = 0; d < D; d++) {
        float s = 0.f;
        for (int j = 0; j < H1; j++) s += r0[j] * w->W0[(size_t)(U + d) * H1 + j];
        dpT[d] = s / (float)T;
      }
      const float* xr = b < valid ? X + (size_t)b * xcols : NULL;
      for (int t = 0; t < T; t++) {
        float dg = 0.f;
        if (xr) for (int d = 0; d < D; d++) dg += dpT[d] * xr[ranges[2] + t * D + d];
        float g = fb.gate[(size_t)b * T + t];
        dgs[(size_t)b * T + t] = dg * (g * (1.0f - g)) * fb.wgt[(size_t)b * T + t];
      }
    }
  }

  /* weight gradients: parallel over output rows, sequential over batch rows */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int i = 0; i < I; i++) {
    float* gr = grads->W0 + (size_t)i * H1;
    for (int j = 0; j < H1; j++) gr[j] = 0.f;
    for (int b = 0; b < B; b++) {
      float a = fb.h0[(size_t)b * I + i];
      const float* r0 = dz0 + (size_t)b * H1;
      for (int j = 0; j < H1; j++) gr[j] += a * r0[j];
    }
  }
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int i = 0; i < H1; i++) {
    float* gr = grads->W1 + (size_t)i * H2;
    for (int j = 0; j < H2; j++) gr[j] = 0.f;
    for (int b = 0; b < B; b++) {
      float a = fb.A0[(size_t)b * H1 + i];
      const float* r1 = dz1 + (size_t)b * H2;
      for (int j = 0; j < H2; j++) gr[j] += a * r1[j];
    }
  }
  for (int i = 0; i < H2; i++) {
    float s = 0.f;
    for (int b = 0; b < B; b++) s += fb.A1[(size_t)b * H2 + i] * dz2[b];
    grads->W2[i] = s;
  }
  if (cfg->kind == ORC_DIN) {
    for (int t = 0; t < T; t++) {
      float s = 0.f;
      for (int b = 0; b < B; b++) s += dgs[(size_t)b * T + t];
      grads->att0[t] = s;
    }
  }
  free(yv); free(dz2); free(dz1); free(dz0); free(dgs);
  free_bufs(&fb);
  return cost;
}

/* gorgonia AdamSolver.Step on one tensor [from memory, SURVEY App. B]:
 *   g += l2*w ; g *= 1/batch ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ;
 *   w += (-eta * m * (1/c1)) / (sqrt(v * (1/c2)) + eps) ; g = 0 */
static void adam_tensor(float* w, float* g, float* m, float* v, size_t n, const orc_adam_cfg* ac,
                        float corr1, float corr2, int batch) {
  const float b1 = (float)ac->beta1, b2 = (float)ac->beta2;
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  const float one_per_batch = 1.0f / (float)batch;
  const float neg_eta = (float)(-ac->lr);
  const float l2 = (float)ac->l2, eps = (float)ac->eps;
  for (size_t i = 0; i < n; i++) {
    float gi = g[i];
    if (ac->adam_l2_before_batch_div) {
      if (l2 != 0.f) gi = gi + w[i] * l2;
      if (ac->adam_div_by_batch && batch > 1) gi = gi * one_per_batch;
    } else {
      if (ac->adam_div_by_batch && batch > 1) gi = gi * one_per_batch;
      if (l2 != 0.f) gi = gi + w[i] * l2;
    }
    float t1 = omb1 * gi;
    float g2 = (gi * gi) * omb2;
    float mi = b1 * m[i] + t1;
    float vi = b2 * v[i] + g2;
    m[i] = mi; v[i] = vi;
    float mhat = mi * corr1;
    float vhat = vi * corr2;
    float den = sqrtf(vhat) + eps;
    w[i] = w[i] + (neg_eta * mhat) / den;
    g[i] = 0.f;
  }
}

void orc_ctr_adam_step(const orc_ctr_cfg* cfg, orc_ctr_weights* w, orc_ctr_weights* grads,
                       orc_adam_state* st, const orc_adam_cfg* ac, int batch) {
  const int I = cfg->U + 2 * cfg->D + cfg->C;
  st->iter++;
  double c1 = 1.0 - pow(ac->beta1, (double)st->iter);
  double c2 = 1.0 - pow(ac->beta2, (double)st->iter);
  float corr1 = 1.0f / (float)c1, corr2 = 1.0f / (float)c2;
  adam_tensor(w->W0, grads->W0, st->m0, st->v0, (size_t)I * cfg->H1, ac, corr1, corr2, batch);
  adam_tensor(w->W1, grads->W1, st->m1, st->v1, (size_t)cfg->H1 * cfg->H2, ac, corr1, corr2, batch);
  adam_tensor(w->W2, grads->W2, st->m2, st->v2, (size_t)cfg->H2, ac, corr1, corr2, batch);
  if (cfg->kind == ORC_DIN)
    adam_tensor(w->att0, grads->att0, st->ma, st->va, (size_t)cfg->T, ac, corr1, corr2, batch);
}

/* model.go:27-213 */
int orc_ctr_train(const orc_ctr_cfg* cfg, orc_ctr_weights* w,
                  const float* X, const float* Y, int64_t rows, int xcols, const int ranges[8],
                  int batch, int epochs, int early_stop, const orc_adam_cfg* ac,
                  int drop_mode, float p0, float p1, uint32_t seed, float* epoch_costs) {
  const int I = cfg->U + 2 * cfg->D + cfg->C;
  const size_t n0 = (size_t)I * cfg->H1, n1 = (size_t)cfg->H1 * cfg->H2, n2 = (size_t)cfg->H2, na = (size_t)cfg->T;
  orc_ctr_weights g;
  g.W0 = (float*)calloc(n0, 4); g.W1 = (float*)calloc(n1, 4); g.W2 = (float*)calloc(n2, 4); g.att0 = (float*)calloc(na, 4);
  orc_adam_state st;
  st.m0 = (float*)calloc(n0, 4); st.v0 = (float*)calloc(n0, 4);
  st.m1 = (float*)calloc(n1, 4); st.v1 = (float*)calloc(n1, 4);
  st.m2 = (float*)calloc(n2, 4); st.v2 = (float*)calloc(n2, 4);
  st.ma = (float*)calloc(na, 4); st.va = (float*)calloc(na, 4);
  st.iter = 0;
  int64_t batches = rows / batch + (rows % batch != 0);
  float best = 3.402823466e+38f; /* math.MaxFloat32 */
  int no_improve = 0, e = 0;
  uint32_t step = 0;
  for (e = 0; e < epochs; e++) {
    float cost = 0.f;
    for (int64_t b = 0; b < batches; b++) {
      int64_t start = b * batch, end = start + batch;
      if (start >= rows) break;
      if (end > rows) end = rows;
      orc_dropout dr;
      memset(&dr, 0, sizeof dr);
      dr.mode = drop_mode ? 2 : 0; dr.p0 = p0; dr.p1 = p1; dr.seed = seed; dr.step = step;
      cost = orc_ctr_loss_grad(cfg, w, X + start * xcols, xcols, ranges, Y + start, batch,
                               (int)(end - start), &dr, &g, NULL);
      orc_ctr_adam_step(cfg, w, &g, &st, ac, batch);
      step++;
    }
    epoch_costs[e] = cost; /* cost of the LAST batch (model.go:198) */
    if (cost < best) { best = cost; no_improve = 0; } else no_improve++;
    if (early_stop != 0 && no_improve >= early_stop) { e++; break; }
  }
  free(g.W0); free(g.W1); free(g.W2); free(g.att0);
  free(st.m0); free(st.v0); free(st.m1); free(st.v1); free(st.m2); free(st.v2); free(st.ma); free(st.va);
  return e;
}

/* model.go:242-352 */
void orc_ctr_predict(const orc_ctr_cfg* cfg, const orc_ctr_weights* w,
                     const float* X, int64_t rows, int xcols, const int ranges[8],
                     int batch, float* y_out) {
  fwd_bufs fb = alloc_bufs(cfg, batch);
  int64_t batches = rows / batch + (rows % batch != 0);
  for (int64_t b = 0; b < batches; b++) {
    int64_t start = b * batch, end = start + batch;
    if (end > rows) end = rows;
    forward_batch(cfg, w, X + start * xcols, xcols, ranges, batch, (int)(end - start), NULL, &fb);
    memcpy(y_out + start, fb.y, sizeof(float) * (size_t)(end - start));
  }
  free_bufs(&fb);
}
