/*
 * orc_ops.c -- oracle (TEST INFRASTRUCTURE, see goctr_oracle.h): op-level restatements of
 * model/activation.go, model/cost.go and the ROC-AUC gate metric.  float32 arithmetic in
 * the order the reference's gorgonia graph evaluates it (one op = one rounded tensor op).
 */
#include "goctr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* model/activation.go:11-16 */
void orc_prelu32(const float* x, float slope, float* out, int n) {
  for (int i = 0; i < n; i++) {
    float ax = fabsf(x[i]);
    float negative = (x[i] - ax) * slope; /* HadamardProd(Sub(x,Abs(x)), slope) */
    float positive = x[i] + ax;           /* Add(x, Abs(x)) */
    out[i] = (negative + positive) * 0.5f;
  }
}

/* model/activation.go:57-83.  Sum(Square(x)) -> Sqrt, Sum(x*y), / (xn*yn + 1e-8) */
int orc_cosine_similarity(const float* x, int Tx, const float* y, int Ty, int B, int D, float* out) {
  if (!(Tx == Ty || Tx == 1 || Ty == 1)) return -1;
  int T = Tx > Ty ? Tx : Ty;
  for (int b = 0; b < B; b++) {
    for (int t = 0; t < T; t++) {
      const float* xr = x + ((size_t)b * Tx + (Tx == 1 ? 0 : t)) * D;
      const float* yr = y + ((size_t)b * Ty + (Ty == 1 ? 0 : t)) * D;
      float sxx = 0.f, syy = 0.f, sxy = 0.f;
      for (int d = 0; d < D; d++) {
        sxx += xr[d] * xr[d];
        syy += yr[d] * yr[d];
        sxy += xr[d] * yr[d];
      }
      float xn = sqrtf(sxx), yn = sqrtf(syy);
      out[(size_t)b * T + t] = sxy / (xn * yn + 1e-8f);
    }
  }
  return 0;
}

/* model/activation.go:23-50 */
int orc_euc_distance(const float* x, int Tx, const float* y, int Ty, int B, int D, float* out) {
  if (!(Tx == Ty || Tx == 1 || Ty == 1)) return -1;
  int T = Tx > Ty ? Tx : Ty;
  for (int b = 0; b < B; b++) {
    for (int t = 0; t < T; t++) {
      const float* xr = x + ((size_t)b * Tx + (Tx == 1 ? 0 : t)) * D;
      const float* yr = y + ((size_t)b * Ty + (Ty == 1 ? 0 : t)) * D;
      float s = 0.f;
      for (int d = 0; d < D; d++) {
        float df = xr[d] - yr[d];
        s += df * df;
      }
      out[(size_t)b * T + t] = sqrtf(s);
    }
  }
  return 0;
}

/* model/cost.go:9-17.  float32(1.0+1e-8) == 1.0f exactly (quirk Q2). */
float orc_bce32(const float* p, const float* y, int n) {
  const float one_eps = (float)(1.0 + 1e-8);
  float s = 0.f;
  for (int i = 0; i < n; i++) {
    float positive = logf(p[i]) * y[i];
    float negative = logf(one_eps - p[i]) * (1.0f - y[i]);
    s += positive + negative;
  }
  return -(s / (float)n);
}

/* model/cost.go:20-23 */
float orc_mse32(const float* p, const float* y, int n) {
  float s = 0.f;
  for (int i = 0; i < n; i++) {
    float d = p[i] - y[i];
    s += d * d;
  }
  return s / (float)n;
}

/* model/cost.go:26-29 */
float orc_rms32(const float* p, const float* y, int n) { return sqrtf(orc_mse32(p, y, n)); }

/* ------------------------------------------------------------------ AUC -- */
typedef struct { double s; double y; } sy_t;
static int cmp_desc(const void* a, const void* b) {
  double x = ((const sy_t*)a)->s, z = ((const sy_t*)b)->s;
  return (x < z) - (x > z);
}

/* nn/metrics/ranking.go:13-57 (binaryClfCurve) + 71-103 (ROCCurve) */
int orc_roc_curve(const double* score, const double* y, double pos_label, int n,
                  double* fpr, double* tpr, double* thr) {
  sy_t* v = (sy_t*)malloc(sizeof(sy_t) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) { v[i].s = score[i]; v[i].y = y[i]; }
  qsort(v, (size_t)n, sizeof(sy_t), cmp_desc);
  int m = 0;
  double tpw = 0., fpw = 0.;
  /* leave slot 0 free for the optional extra threshold position */
  for (int i = 0; i < n;) {
    int j = i;
    while (j < n && !(v[j].s < v[i].s)) { /* tie group */
      if (v[j].y == pos_label) tpw += 1.; else fpw += 1.;
      j++;
    }
    tpr[m + 1] = tpw; fpr[m + 1] = fpw; thr[m + 1] = v[i].s;
    m++;
    i = j;
  }
  int off = 1, cnt = m;
  if (m == 0 || fpr[1] != 0.) { /* ranking.go:74-79 */
    fpr[0] = 0.; tpr[0] = 0.; thr[0] = (m ? thr[1] : 0.) + 1.;
    off = 0; cnt = m + 1;
  }
  if (off) { memmove(fpr, fpr + 1, sizeof(double) * (size_t)m); memmove(tpr, tpr + 1, sizeof(double) * (size_t)m);
             memmove(thr, thr + 1, sizeof(double) * (size_t)m); }
  double fpmax = cnt ? fpr[cnt - 1] : 0., tpmax = cnt ? tpr[cnt - 1] : 0.;
  for (int i = 0; i < cnt; i++) {
    fpr[i] = fpmax <= 0. ? NAN : fpr[i] * (1. / fpmax);
    tpr[i] = tpmax <= 0. ? NAN : tpr[i] * (1. / tpmax);
  }
  free(v);
  return cnt;
}

/* ranking.go:106-118 (AUC trapezoid) via 144-150 (ROCAUCScore); labels thresholded at .5
 * by the callers utils/util.go:116-148 */
double orc_roc_auc(const double* score, const double* y, int n) {
  double* buf = (double*)malloc(sizeof(double) * 3 * (size_t)(n + 2));
  double* yb = (double*)malloc(sizeof(double) * (size_t)(n + 1));
  for (int i = 0; i < n; i++) yb[i] = y[i] > 0.5 ? 1.0 : 0.0;
  double *fpr = buf, *tpr = buf + (n + 2), *thr = buf + 2 * (n + 2);
  int m = orc_roc_curve(score, yb, 1.0, n, fpr, tpr, thr);
  double auc = 0., xp = 0., yp = 0.;
  for (int i = 0; i < m; i++) {
    auc += (fpr[i] - xp) * (tpr[i] + yp) / 2.;
    xp = fpr[i]; yp = tpr[i];
  }
  free(buf); free(yb);
  return auc;
}

float orc_roc_auc32(const float* score, const float* y, int n) {
  double* s = (double*)malloc(sizeof(double) * (size_t)(n + 1));
  double* t = (double*)malloc(sizeof(double) * (size_t)(n + 1));
  for (int i = 0; i < n; i++) { s[i] = (double)score[i]; t[i] = (double)y[i]; }
  float r = (float)orc_roc_auc(s, t, n);
  free(s); free(t);
  return r;
}
