/* TEST INFRASTRUCTURE ONLY - CPU oracle, never linked into or called by the product path.
 *
 * Oracle for the TRAINABLE-EMBEDDING EXTENSION (SURVEY F3, K18, 8(e) "Trainable embeddings (extension, cfg4)";
 * BASELINE north_star "embedding gather and SGD scatter-add").  The reference keeps the item embeddings frozen while
 * DIN / YouTube-DNN train (model/din/din.go:161-169, model/youtube/dnn.go:152-154 list the learnables), so there is NO
 * reference code to restate: PARITY UNPINNED.  What is pinned instead: the forward below is the float64 evaluation of
 * the same graph as orc_ctr.c's fwd_row (din.go:219-323, dnn.go:162-184, activation.go:23-83, cost.go:9-17), and the
 * analytic gradient with respect to the embedding rows is checked against central finite differences of this very
 * loss (tests/test_oracle_embtrain.py).
 *
 *   x_t = E[ub_ids[b,t]] (zero row for ids outside [0,V)),  v = E[item_ids[b]]
 *   DIN cosine : s = x.v, den = |x||v| + 1e-8, w = (s/den + 1)/2 ; DIN euclid: w = 1 - |x - v| ; g = sigm(w att0[t])
 *   YouTube    : g = 1
 *   p = (1/T) sum_t g_t x_t ; h0 = [u | p | v | c] ; three sigmoid layers (dropout factors k = mask/keep) ; BCE mean over B
 *
 *   dx_t = (g_t/T) dp + q_t dw_t/dx_t ,  dv = dh0[item segment] + sum_t q_t dw_t/dv ,
 *   q_t = ((dp . x_t)/T) g_t (1 - g_t) att0[t]
 *   cosine: dw/dx = (v/den - s |v| x / (|x| den^2)) / 2 , dw/dv = (x/den - s |x| v / (|v| den^2)) / 2   (terms with a
 *           zero norm in the denominator are dropped: d|x|/dx := 0 at x = 0)
 *   euclid: dw/dx = -(x - v)/|x - v| , dw/dv = (x - v)/|x - v|   (:= 0 at x = v)
 * Padded batch rows (b >= valid) have no ids and take part in the loss only (model.go:357-371).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "goctr_oracle.h"

static double sigm64(double x) {      /* the clamps of gorgonia's float32 sigmoid, evaluated in float64 */
  if (x < -88.0) return 0.0;
  if (x > 15.0) return 1.0;
  return 1.0 / (1.0 + exp(-x));
}

double orc_embtrain_loss_grad(const orc_ctr_cfg* cfg, const orc_ctr_weights* w, const double* E, int64_t V,
                              const int32_t* ub_ids, const int32_t* item_ids, const float* ufeat, const float* cfeat,
                              const float* Y, int B, int valid, const orc_dropout* drop, double* dE) {
  const int U = cfg->U, T = cfg->T, D = cfg->D, C = cfg->C, H1 = cfg->H1, H2 = cfg->H2;
  const int I = U + 2 * D + C;
  const int din = cfg->kind == ORC_DIN, cosine = cfg->att == ORC_ATT_COSINE;
  if (dE) memset(dE, 0, sizeof(double) * (size_t)V * (size_t)D);
  double* h0 = (double*)malloc(sizeof(double) * (size_t)I);
  double* x = (double*)calloc((size_t)T * (size_t)D, sizeof(double));
  double* g = (double*)malloc(sizeof(double) * (size_t)T);
  double* wv = (double*)malloc(sizeof(double) * (size_t)T);
  double* P0 = (double*)malloc(sizeof(double) * (size_t)H1);
  double* K0 = (double*)malloc(sizeof(double) * (size_t)H1);
  double* A0 = (double*)malloc(sizeof(double) * (size_t)H1);
  double* P1 = (double*)malloc(sizeof(double) * (size_t)H2);
  double* K1 = (double*)malloc(sizeof(double) * (size_t)H2);
  double* A1 = (double*)malloc(sizeof(double) * (size_t)H2);
  double* dz1 = (double*)malloc(sizeof(double) * (size_t)H2);
  double* dz0 = (double*)malloc(sizeof(double) * (size_t)H1);
  double* dh = (double*)malloc(sizeof(double) * (size_t)(2 * D));
  double loss = 0.0;
  for (int b = 0; b < B; b++) {
    const int live = b < valid;
    memset(h0, 0, sizeof(double) * (size_t)I);
    memset(x, 0, sizeof(double) * (size_t)T * (size_t)D);
    int64_t item = -1;
    if (live) {
      for (int j = 0; j < U; j++) h0[j] = ufeat[(size_t)b * U + j];
      for (int j = 0; j < C; j++) h0[U + 2 * D + j] = cfeat[(size_t)b * C + j];
      item = item_ids[b];
      if (item >= 0 && item < V) for (int d = 0; d < D; d++) h0[U + D + d] = E[item * D + d];
      else item = -1;
      for (int t = 0; t < T; t++) {
        const int64_t id = ub_ids[(size_t)b * T + t];
        if (id >= 0 && id < V) for (int d = 0; d < D; d++) x[t * D + d] = E[id * D + d];
      }
    }
    const double* v = h0 + U + D;
    double nv = 0;
    for (int d = 0; d < D; d++) nv += v[d] * v[d];
    nv = sqrt(nv);
    for (int t = 0; t < T; t++) {
      const double* xt = x + t * D;
      if (din) {
        if (cosine) {
          double sxx = 0, sxy = 0;
          for (int d = 0; d < D; d++) { sxx += xt[d] * xt[d]; sxy += xt[d] * v[d]; }
          wv[t] = (sxy / (sqrt(sxx) * nv + 1e-8) + 1.0) / 2.0;
        } else {
          double s = 0;
          for (int d = 0; d < D; d++) s += (xt[d] - v[d]) * (xt[d] - v[d]);
          wv[t] = 1.0 - sqrt(s);
        }
        g[t] = sigm64(wv[t] * (double)w->att0[t]);
      } else { g[t] = 1.0; wv[t] = 0.0; }
      for (int d = 0; d < D; d++) h0[U + d] += g[t] * xt[d];
    }
    for (int d = 0; d < D; d++) h0[U + d] /= (double)T;
    for (int j = 0; j < H1; j++) {
      double z = 0;
      for (int i = 0; i < I; i++) z += h0[i] * (double)w->W0[(size_t)i * H1 + j];
      P0[j] = sigm64(z); K0[j] = 1.0;
      if (drop && drop->mode && drop->p0 > 0.f) {
        const double m = drop->mode == 1 ? drop->m0[(size_t)b * H1 + j]
                                         : orc_dropout_keep(drop->seed, drop->step, 0, (uint32_t)b, (uint32_t)j, drop->p0);
        K0[j] = m / (1.0 - (double)drop->p0);
      }
      A0[j] = P0[j] * K0[j];
    }
    for (int j = 0; j < H2; j++) {
      double z = 0;
      for (int i = 0; i < H1; i++) z += A0[i] * (double)w->W1[(size_t)i * H2 + j];
      P1[j] = sigm64(z); K1[j] = 1.0;
      if (drop && drop->mode && drop->p1 > 0.f) {
        const double m = drop->mode == 1 ? drop->m1[(size_t)b * H2 + j]
                                         : orc_dropout_keep(drop->seed, drop->step, 1, (uint32_t)b, (uint32_t)j, drop->p1);
        K1[j] = m / (1.0 - (double)drop->p1);
      }
      A1[j] = P1[j] * K1[j];
    }
    double z2 = 0;
    for (int i = 0; i < H2; i++) z2 += A1[i] * (double)w->W2[i];
    const double yh = sigm64(z2);
    const double y = live && Y ? (double)Y[b] : 0.0;
    loss += -(y * log(yh) + (1.0 - y) * log(1.0 - yh));          /* cost.go:9-17 (c = 1.0f exactly) */
    if (!dE || !live) continue;
    const double dz2 = (yh - y) / (double)B;
    for (int j = 0; j < H2; j++) dz1[j] = dz2 * (double)w->W2[j] * K1[j] * P1[j] * (1.0 - P1[j]);
    for (int i = 0; i < H1; i++) {
      double s = 0;
      for (int j = 0; j < H2; j++) s += dz1[j] * (double)w->W1[(size_t)i * H2 + j];
      dz0[i] = s * K0[i] * P0[i] * (1.0 - P0[i]);
    }
    for (int n = 0; n < 2 * D; n++) {                              /* d cost / d [p | v] */
      double s = 0;
      for (int j = 0; j < H1; j++) s += dz0[j] * (double)w->W0[(size_t)(U + n) * H1 + j];
      dh[n] = s;
    }
    const double* dp = dh;
    double* dv = dh + D;                                           /* grows by the attention terms below */
    for (int t = 0; t < T; t++) {
      const int64_t id = ub_ids[(size_t)b * T + t];
      if (id < 0 || id >= V) continue;                             /* empty slot: x = 0 is a constant, and q_t = 0 */
      const double* xt = x + t * D;
      double* dx = dE + id * D;
      for (int d = 0; d < D; d++) dx[d] += g[t] / (double)T * dp[d];
      if (!din) continue;
      double dot = 0;
      for (int d = 0; d < D; d++) dot += dp[d] * xt[d];
      const double q = dot / (double)T * g[t] * (1.0 - g[t]) * (double)w->att0[t];
      if (cosine) {
        double sxx = 0, sxy = 0;
        for (int d = 0; d < D; d++) { sxx += xt[d] * xt[d]; sxy += xt[d] * v[d]; }
        const double nx = sqrt(sxx), den = nx * nv + 1e-8;
        const double cx = nx > 0 ? sxy * nv / (nx * den * den) : 0.0;
        const double cv = nv > 0 ? sxy * nx / (nv * den * den) : 0.0;
        for (int d = 0; d < D; d++) {
          dx[d] += q * 0.5 * (v[d] / den - cx * xt[d]);
          dv[d] += q * 0.5 * (xt[d] / den - cv * v[d]);
        }
      } else {
        double s = 0;
        for (int d = 0; d < D; d++) s += (xt[d] - v[d]) * (xt[d] - v[d]);
        const double r = sqrt(s);
        if (r > 0)
          for (int d = 0; d < D; d++) {
            dx[d] += q * -(xt[d] - v[d]) / r;
            dv[d] += q * (xt[d] - v[d]) / r;
          }
      }
    }
    if (item >= 0) for (int d = 0; d < D; d++) dE[item * D + d] += dv[d];
  }
  free(h0); free(x); free(g); free(wv); free(P0); free(K0); free(A0); free(P1); free(K1); free(A1);
  free(dz1); free(dz0); free(dh);
  return loss / (double)B;
}
