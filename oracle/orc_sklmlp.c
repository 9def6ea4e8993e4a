/*
 * orc_sklmlp.c -- oracle (TEST INFRASTRUCTURE, see goctr_oracle.h): float64 restatement of the
 * sklearn-port MLP used by model/mlp (nn/neural_network/basemlp64.go).
 *
 * Packed parameter layout (basemlp64.go:432-463): for each layer i,
 *   [ intercepts_i (fanOut) | coefs_i (fanIn x fanOut, row-major) ].
 * Quirks kept on purpose: Q7 per-parameter Adam beta powers, Q9 tanh(-z),
 * Q10 max-abs "batch normalisation" (deltas divided by M even when M == 0), Q12 relu' tests a==0.
 * Q11 (stale rows in a short last batch) is avoided by requiring n % batch == 0 in orc_mlp_fit.
 * GEMM summation order: gonum's blocked Dgemm order is not reproduced (plain k-ordered sums);
 * compare at 1e-9 relative, never bit-exact.
 */
#include "goctr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

size_t orc_mlp_nparams(const orc_mlp_cfg* cfg) {
  size_t n = 0;
  for (int i = 0; i < cfg->n_layers - 1; i++) n += (size_t)(1 + cfg->units[i]) * cfg->units[i + 1];
  return n;
}

static void layer_ptrs(const orc_mlp_cfg* cfg, double* theta, double** b, double** W) {
  size_t off = 0;
  for (int i = 0; i < cfg->n_layers - 1; i++) {
    b[i] = theta + off; off += (size_t)cfg->units[i + 1];
    W[i] = theta + off; off += (size_t)cfg->units[i] * cfg->units[i + 1];
  }
}

/* basemlp64.go:79-117 */
static void act_inplace(int kind, double* z, size_t n) {
  switch (kind) {
    case ORC_ACT_IDENTITY: break;
    case ORC_ACT_LOGISTIC: for (size_t i = 0; i < n; i++) z[i] = 1 / (1 + exp(-z[i])); break;
    case ORC_ACT_TANH: for (size_t i = 0; i < n; i++) z[i] = tanh(-z[i]); break; /* Q9 */
    case ORC_ACT_RELU: for (size_t i = 0; i < n; i++) if (z[i] < 0) z[i] = 0; break;
  }
}
/* basemlp64.go:120-148 */
static void deriv_inplace(int kind, const double* a, double* d, size_t n) {
  switch (kind) {
    case ORC_ACT_IDENTITY: break;
    case ORC_ACT_LOGISTIC: for (size_t i = 0; i < n; i++) d[i] *= a[i] * (1 - a[i]); break;
    case ORC_ACT_TANH: for (size_t i = 0; i < n; i++) d[i] *= 1 - a[i] * a[i]; break;
    case ORC_ACT_RELU: for (size_t i = 0; i < n; i++) if (a[i] == 0) d[i] = 0; break; /* Q12 */
  }
}

/* activations[i+1] = act(activations[i] . W_i + b_i)   basemlp64.go:259-274 */
static void forward(const orc_mlp_cfg* cfg, double* const* b, double* const* W, double** acts, int n) {
  const int L = cfg->n_layers;
#pragma omp parallel for num_threads(orc_get_threads()) schedule(static)
  for (int r = 0; r < n; r++) {
    for (int i = 0; i < L - 1; i++) {
      const int fi = cfg->units[i], fo = cfg->units[i + 1];
      const double* a = acts[i] + (size_t)r * fi;
      double* z = acts[i + 1] + (size_t)r * fo;
      for (int j = 0; j < fo; j++) z[j] = 0;
      for (int k = 0; k < fi; k++) {
        const double av = a[k];
        const double* wr = W[i] + (size_t)k * fo;
        for (int j = 0; j < fo; j++) z[j] += av * wr[j];
      }
      for (int j = 0; j < fo; j++) z[j] += b[i][j]; /* addIntercepts64 :205 */
      if (i + 1 != L - 1) act_inplace(cfg->activation, z, (size_t)fo);
      else act_inplace(ORC_ACT_LOGISTIC, z, (size_t)fo); /* binary classifier output :270-273 */
    }
  }
}

void orc_mlp_predict(const orc_mlp_cfg* cfg, const double* theta, const double* X, int n, double* out) {
  const int L = cfg->n_layers;
  double *b[8], *W[8], *acts[8];
  layer_ptrs(cfg, (double*)theta, b, W);
  acts[0] = (double*)X;
  for (int i = 1; i < L; i++) acts[i] = i == L - 1 ? out : (double*)malloc(sizeof(double) * (size_t)n * cfg->units[i]);
  forward(cfg, b, W, acts, n);
  for (int i = 1; i < L - 1; i++) free(acts[i]);
}

/* basemlp64.go:340-406 */
double orc_mlp_loss_grad(const orc_mlp_cfg* cfg, double* theta, const double* X, const double* Y,
                         int n, double* grads) {
  const int L = cfg->n_layers;
  const size_t np = orc_mlp_nparams(cfg);
  double *b[8], *W[8], *gb[8], *gW[8], *acts[8], *deltas[8], *bn[8];
  if (cfg->weight_decay > 0) /* :342-346 */
    for (size_t i = 0; i < np; i++) theta[i] *= (1 - cfg->weight_decay);
  layer_ptrs(cfg, theta, b, W);
  layer_ptrs(cfg, grads, gb, gW);
  acts[0] = (double*)X;
  for (int i = 1; i < L; i++) {
    acts[i] = (double*)malloc(sizeof(double) * (size_t)n * cfg->units[i]);
    deltas[i - 1] = (double*)malloc(sizeof(double) * (size_t)n * cfg->units[i]);
    bn[i - 1] = (double*)calloc((size_t)cfg->units[i], sizeof(double));
  }
  forward(cfg, b, W, acts, n);
  if (cfg->batch_normalize) { /* :277-299 */
    for (int i = 0; i < L - 2; i++) {
      const int fo = cfg->units[i + 1];
      double* a = acts[i + 1];
      for (int o = 0; o < fo; o++) {
        double M = 0;
        for (int r = 0; r < n; r++) { double v = fabs(a[(size_t)r * fo + o]); if (M < v) M = v; }
        if (M > 0) for (int r = 0; r < n; r++) a[(size_t)r * fo + o] /= M;
        bn[i][o] = M;
      }
    }
  }
  /* binary_log_loss :180-195 */
  const int no = cfg->units[L - 1];
  const double hmin = nextafter(0.0, 1.0), hmax = nextafter(1.0, 0.0);
  double sum = 0;
  const double* H = acts[L - 1];
  for (size_t i = 0; i < (size_t)n * no; i++) {
    double h = H[i];
    if (h < hmin) h = hmin; else if (h > hmax) h = hmax;
    sum += -Y[i] * log(h) - (1 - Y[i]) * log1p(-h);
  }
  double loss = sum / (double)n;
  double s2 = 0; /* sumCoefSquares :310-318 */
  for (int i = 0; i < L - 1; i++) {
    size_t cnt = (size_t)cfg->units[i] * cfg->units[i + 1];
    for (size_t k = 0; k < cnt; k++) s2 += W[i][k] * W[i][k];
  }
  loss += (0.5 * cfg->alpha) * s2 / (double)n;

  const int last = L - 2;
  for (size_t i = 0; i < (size_t)n * no; i++) deltas[last][i] = H[i] - Y[i]; /* :373-381 */

  for (int layer = last; layer >= 0; layer--) {
    const int fi = cfg->units[layer], fo = cfg->units[layer + 1];
    /* computeLossGrad :322-330: coefGrads = a^T.delta / n + alpha/n * W ; interceptGrads = mean */
    const double inv = 1 / (double)n;
#pragma omp parallel for num_threads(orc_get_threads()) schedule(static)
    for (int k = 0; k < fi; k++) {
      double* g = gW[layer] + (size_t)k * fo;
      for (int j = 0; j < fo; j++) g[j] = 0;
      for (int r = 0; r < n; r++) {
        const double av = acts[layer][(size_t)r * fi + k];
        const double* d = deltas[layer] + (size_t)r * fo;
        for (int j = 0; j < fo; j++) g[j] += av * d[j];
      }
      for (int j = 0; j < fo; j++) g[j] = inv * g[j];
      for (int j = 0; j < fo; j++) g[j] += (cfg->alpha / (double)n) * W[layer][(size_t)k * fo + j];
    }
    for (int j = 0; j < fo; j++) { /* matRowMean64 :213-226 */
      double s = 0;
      for (int r = 0; r < n; r++) s += deltas[layer][(size_t)r * fo + j];
      gb[layer][j] = s / (double)n;
    }
    if (layer >= 1) { /* :386-398 */
      double* dprev = deltas[layer - 1];
#pragma omp parallel for num_threads(orc_get_threads()) schedule(static)
      for (int r = 0; r < n; r++) {
        const double* d = deltas[layer] + (size_t)r * fo;
        for (int k = 0; k < fi; k++) {
          double s = 0;
          const double* wr = W[layer] + (size_t)k * fo;
          for (int j = 0; j < fo; j++) s += d[j] * wr[j];
          dprev[(size_t)r * fi + k] = s;
        }
      }
      deriv_inplace(cfg->activation, acts[layer], dprev, (size_t)n * fi);
      if (cfg->batch_normalize) /* :302-308, Q10: unconditional divide */
        for (int r = 0; r < n; r++)
          for (int o = 0; o < fi; o++) dprev[(size_t)r * fi + o] /= bn[layer - 1][o];
    }
  }
  for (int i = 1; i < L; i++) { free(acts[i]); free(deltas[i - 1]); free(bn[i - 1]); }
  return loss;
}

void orc_mlp_opt_init(orc_mlp_opt* o, int solver, size_t np) {
  memset(o, 0, sizeof *o);
  o->solver = solver;
  o->lr_init = 0.001; o->beta1 = 0.9; o->beta2 = 0.999; o->eps = 1e-8; /* :228-254 */
  o->momentum = 0.9; o->nesterov = 1;
  o->ms = (double*)calloc(np, sizeof(double));
  o->vs = (double*)calloc(np, sizeof(double));
  o->velocities = (double*)calloc(np, sizeof(double));
  o->beta1t = 1; o->beta2t = 1; o->t = 0; o->lr = o->lr_init;
}
void orc_mlp_opt_free(orc_mlp_opt* o) { free(o->ms); free(o->vs); free(o->velocities); }

void orc_mlp_update(orc_mlp_opt* o, double* theta, const double* grads, size_t np) {
  if (o->solver == ORC_SOLVER_ADAM) { /* :1075-1091 */
    o->t += 1;
    for (size_t i = 0; i < np; i++) {
      const double g = grads[i];
      o->ms[i] = o->beta1 * o->ms[i] + (1 - o->beta1) * g;
      o->vs[i] = o->beta2 * o->vs[i] + (1 - o->beta2) * g * g;
      o->beta1t *= o->beta1; /* Q7: advanced once per PARAMETER */
      o->beta2t *= o->beta2;
      o->lr = o->lr_init * sqrt(1 - o->beta2t) / (1. - o->beta1t);
      theta[i] += -o->lr * o->ms[i] / (sqrt(o->vs[i]) + o->eps);
    }
  } else { /* :1024-1039 */
    if (o->t == 0) o->lr = o->lr_init;
    o->t += 1;
    for (size_t i = 0; i < np; i++) {
      double update = o->momentum * o->velocities[i] - o->lr * grads[i];
      o->velocities[i] = update;
      if (o->nesterov) theta[i] += o->momentum * update - o->lr * grads[i];
      else theta[i] += update;
    }
  }
}

/* fitStochastic :729-857 with a given row order */
int orc_mlp_fit(const orc_mlp_cfg* cfg, double* theta, orc_mlp_opt* opt,
                const double* X, const double* Y, int64_t n, int batch, int max_iter,
                double tol, int n_iter_no_change, const int32_t* perm, double* loss_curve) {
  const size_t np = orc_mlp_nparams(cfg);
  const int F = cfg->units[0], no = cfg->units[cfg->n_layers - 1];
  double* grads = (double*)malloc(sizeof(double) * np);
  double* Xb = (double*)malloc(sizeof(double) * (size_t)batch * F);
  double* Yb = (double*)malloc(sizeof(double) * (size_t)batch * no);
  double best = INFINITY;
  int no_improve = 0, it = 0;
  for (it = 0; it < max_iter; it++) {
    double acc = 0;
    for (int64_t s = 0; s + batch <= n; s += batch) {
      const double *xb = X + s * F, *yb = Y + s * no;
      if (perm) {
        const int32_t* p = perm + (int64_t)it * n + s;
        for (int r = 0; r < batch; r++) {
          memcpy(Xb + (size_t)r * F, X + (int64_t)p[r] * F, sizeof(double) * (size_t)F);
          memcpy(Yb + (size_t)r * no, Y + (int64_t)p[r] * no, sizeof(double) * (size_t)no);
        }
        xb = Xb; yb = Yb;
      }
      double bl = orc_mlp_loss_grad(cfg, theta, xb, yb, batch, grads);
      acc += bl * (double)batch;
      orc_mlp_update(opt, theta, grads, np);
    }
    double loss = acc / (double)n;
    loss_curve[it] = loss;
    /* updateNoImprovementCount :859-895 (no early stopping) */
    if (loss > best - tol) no_improve++; else no_improve = 0;
    if (loss < best) best = loss;
    if (no_improve > n_iter_no_change) { it++; break; } /* constant schedule => stop :826-835 */
  }
  free(grads); free(Xb); free(Yb);
  return it;
}
