/*
 * orc_sklmlp.c -- oracle (TEST INFRASTRUCTURE, see goctr_oracle.h): float64 restatement of the
 * sklearn-port MLP used by model/mlp (nn/neural_network/basemlp64.go).
 *
 * Packed parameter layout (basemlp64.go:432-463): for each layer i,
 *   [ intercepts_i (fanOut) | coefs_i (fanIn x fanOut, row-major) ].
 * Quirks kept on purpose: Q7 per-parameter Adam beta powers, Q9 tanh(-z),
 * Q10 max-abs "batch normalisation" (deltas divided by M even when M == 0), Q12 relu' tests a==0.
 * Q11 (stale rows in a short last batch, basemlp64.go:790-812) is REPRODUCED: orc_mlp_loss_grad_rows / orc_mlp_fit.
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

/* activations[i+1] = act(activations[i] . W_i + b_i)   basemlp64.go:259-274, with the reference's ROW COUNTS: the GEMM of
 * layer i takes its row count from activations[i] (gonum blas64.Gemm reads m, k from A and n from B; C's Rows field is
 * ignored), addIntercepts64 / the activation loop over activations[i+1].Rows.  In a full batch every block has B rows.  In
 * the SHORT LAST BATCH of an epoch (quirk Q11, basemlp64.go:790-802) only activations[0] = Xbatch has ns < B rows -- the
 * loop `for _, a := range activations { a.Rows = Xbatch.Rows }` at :800-802 mutates a COPY of each header -- so:
 *   layer 0: rows [0, ns) of activations[1] are overwritten by the product (beta = 0), rows [ns, B) keep the PREVIOUS
 *            batch's post-activation values; then + b_0 and the activation run over all B rows (the stale rows become
 *            act(stale + b_0));
 *   layers >= 1: B rows in, B rows out (stale rows included). */
static void forward_rows(const orc_mlp_cfg* cfg, double* const* b, double* const* W, double** acts, int ns, int B) {
  const int L = cfg->n_layers;
  for (int i = 0; i < L - 1; i++) {
    const int fi = cfg->units[i], fo = cfg->units[i + 1];
    const int m = i == 0 ? ns : B;            /* activations[i].Rows */
    const int kind = i + 1 != L - 1 ? cfg->activation : ORC_ACT_LOGISTIC; /* binary classifier output :270-273 */
#pragma omp parallel for num_threads(orc_get_threads()) schedule(static)
    for (int r = 0; r < B; r++) {
      double* z = acts[i + 1] + (size_t)r * fo;
      if (r < m) {
        const double* a = acts[i] + (size_t)r * fi;
        for (int j = 0; j < fo; j++) z[j] = 0;
        for (int k = 0; k < fi; k++) {
          const double av = a[k];
          const double* wr = W[i] + (size_t)k * fo;
          for (int j = 0; j < fo; j++) z[j] += av * wr[j];
        }
      }
      for (int j = 0; j < fo; j++) z[j] += b[i][j]; /* addIntercepts64 :205 (a.Rows = B rows) */
      act_inplace(kind, z, (size_t)fo);
    }
  }
}

void orc_mlp_predict(const orc_mlp_cfg* cfg, const double* theta, const double* X, int n, double* out) {
  const int L = cfg->n_layers;
  double *b[8], *W[8], *acts[8];
  layer_ptrs(cfg, (double*)theta, b, W);
  acts[0] = (double*)X;
  for (int i = 1; i < L; i++) acts[i] = i == L - 1 ? out : (double*)malloc(sizeof(double) * (size_t)n * cfg->units[i]);
  forward_rows(cfg, b, W, acts, n, n);
  for (int i = 1; i < L - 1; i++) free(acts[i]);
}

/* basemlp64.go:340-406 on the CALLER's activation / delta blocks (B rows each, as fit allocates them at :529-545), for a
 * batch of ns <= B rows.  ns == B is the ordinary batch.  ns < B is the short last batch (Q11), literally:
 *   nSamples = X.Rows = ns (:341)
 *   loss     = sum over y.Rows = ns rows / h.Rows = B (:180-195: `sum / float64(h.Rows)`)  +  0.5 alpha |W|^2 / ns (:361)
 *   deltas[last] rows [0, ns) = h - y (:373-381 loops y.Rows); rows [ns, B) keep the previous batch's values
 *   computeLossGrad(layer): coefGrads = (1/ns) A^T . delta with k = activations[layer].Rows (ns for layer 0, B above),
 *                           + alpha/ns W;  interceptGrads = matRowMean64(deltas[layer]) over deltas.Rows = B, / B (:213-226)
 *   deltas[i-1] = deltas[i] . W_i^T over deltas[i].Rows = B rows; derivative loop over activations[i].Rows = B rows. */
double orc_mlp_loss_grad_rows(const orc_mlp_cfg* cfg, double* theta, const double* X, const double* Y, int ns, int B,
                              double* const* acts_in, double* const* deltas, double* grads) {
  const int L = cfg->n_layers;
  const size_t np = orc_mlp_nparams(cfg);
  double *b[8], *W[8], *gb[8], *gW[8], *acts[8], *bn[8];
  if (cfg->weight_decay > 0) /* :342-346 */
    for (size_t i = 0; i < np; i++) theta[i] *= (1 - cfg->weight_decay);
  layer_ptrs(cfg, theta, b, W);
  layer_ptrs(cfg, grads, gb, gW);
  acts[0] = (double*)X;
  for (int i = 1; i < L; i++) {
    acts[i] = acts_in[i];
    bn[i - 1] = (double*)calloc((size_t)cfg->units[i], sizeof(double));
  }
  forward_rows(cfg, b, W, acts, ns, B);
  if (cfg->batch_normalize) { /* :277-299 (activation.Rows = B) */
    for (int i = 0; i < L - 2; i++) {
      const int fo = cfg->units[i + 1];
      double* a = acts[i + 1];
      for (int o = 0; o < fo; o++) {
        double M = 0;
        for (int r = 0; r < B; r++) { double v = fabs(a[(size_t)r * fo + o]); if (M < v) M = v; }
        if (M > 0) for (int r = 0; r < B; r++) a[(size_t)r * fo + o] /= M;
        bn[i][o] = M;
      }
    }
  }
  /* binary_log_loss :180-195 */
  const int no = cfg->units[L - 1];
  const double hmin = nextafter(0.0, 1.0), hmax = nextafter(1.0, 0.0);
  double sum = 0;
  const double* H = acts[L - 1];
  for (size_t i = 0; i < (size_t)ns * no; i++) {
    double h = H[i];
    if (h < hmin) h = hmin; else if (h > hmax) h = hmax;
    sum += -Y[i] * log(h) - (1 - Y[i]) * log1p(-h);
  }
  double loss = sum / (double)B;
  double s2 = 0; /* sumCoefSquares :310-318 */
  for (int i = 0; i < L - 1; i++) {
    size_t cnt = (size_t)cfg->units[i] * cfg->units[i + 1];
    for (size_t k = 0; k < cnt; k++) s2 += W[i][k] * W[i][k];
  }
  loss += (0.5 * cfg->alpha) * s2 / (double)ns;

  const int last = L - 2;
  for (size_t i = 0; i < (size_t)ns * no; i++) deltas[last][i] = H[i] - Y[i]; /* :373-381 */

  for (int layer = last; layer >= 0; layer--) {
    const int fi = cfg->units[layer], fo = cfg->units[layer + 1];
    const int kr = layer == 0 ? ns : B;       /* activations[layer].Rows */
    /* computeLossGrad :322-330: coefGrads = a^T.delta / n + alpha/n * W ; interceptGrads = mean */
    const double inv = 1 / (double)ns;
#pragma omp parallel for num_threads(orc_get_threads()) schedule(static)
    for (int k = 0; k < fi; k++) {
      double* g = gW[layer] + (size_t)k * fo;
      for (int j = 0; j < fo; j++) g[j] = 0;
      for (int r = 0; r < kr; r++) {
        const double av = acts[layer][(size_t)r * fi + k];
        const double* d = deltas[layer] + (size_t)r * fo;
        for (int j = 0; j < fo; j++) g[j] += av * d[j];
      }
      for (int j = 0; j < fo; j++) g[j] = inv * g[j];
      for (int j = 0; j < fo; j++) g[j] += (cfg->alpha / (double)ns) * W[layer][(size_t)k * fo + j];
    }
    for (int j = 0; j < fo; j++) { /* matRowMean64 :213-226 (a.Rows = B) */
      double s = 0;
      for (int r = 0; r < B; r++) s += deltas[layer][(size_t)r * fo + j];
      gb[layer][j] = s / (double)B;
    }
    if (layer >= 1) { /* :386-398 */
      double* dprev = deltas[layer - 1];
#pragma omp parallel for num_threads(orc_get_threads()) schedule(static)
      for (int r = 0; r < B; r++) {
        const double* d = deltas[layer] + (size_t)r * fo;
        for (int k = 0; k < fi; k++) {
          double s = 0;
          const double* wr = W[layer] + (size_t)k * fo;
          for (int j = 0; j < fo; j++) s += d[j] * wr[j];
          dprev[(size_t)r * fi + k] = s;
        }
      }
      deriv_inplace(cfg->activation, acts[layer], dprev, (size_t)B * fi);
      if (cfg->batch_normalize) /* :302-308, Q10: unconditional divide */
        for (int r = 0; r < B; r++)
          for (int o = 0; o < fi; o++) dprev[(size_t)r * fi + o] /= bn[layer - 1][o];
    }
  }
  for (int i = 1; i < L; i++) free(bn[i - 1]);
  return loss;
}

/* one batch of n rows on fresh blocks (what every batch but a short last one is) */
double orc_mlp_loss_grad(const orc_mlp_cfg* cfg, double* theta, const double* X, const double* Y,
                         int n, double* grads) {
  const int L = cfg->n_layers;
  double *acts[8] = {0}, *deltas[8] = {0};
  for (int i = 1; i < L; i++) {
    acts[i] = (double*)malloc(sizeof(double) * (size_t)n * cfg->units[i]);
    deltas[i - 1] = (double*)malloc(sizeof(double) * (size_t)n * cfg->units[i]);
  }
  const double loss = orc_mlp_loss_grad_rows(cfg, theta, X, Y, n, n, acts, deltas, grads);
  for (int i = 1; i < L; i++) { free(acts[i]); free(deltas[i - 1]); }
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

/* fitStochastic :729-857 with a given row order.  The activation / delta blocks live across the batches of the call
 * like the reference's (fit :529-545: one allocation of BatchSize rows per layer, zero-filled by `make`), which is what
 * makes the short last batch of an epoch see the previous batch's rows (Q11; orc_mlp_loss_grad_rows). */
int orc_mlp_fit(const orc_mlp_cfg* cfg, double* theta, orc_mlp_opt* opt,
                const double* X, const double* Y, int64_t n, int batch, int max_iter,
                double tol, int n_iter_no_change, const int32_t* perm, double* loss_curve) {
  const size_t np = orc_mlp_nparams(cfg);
  const int L = cfg->n_layers;
  const int F = cfg->units[0], no = cfg->units[L - 1];
  if (batch > n) batch = (int)n;             /* :517-520 "Got batchsize larger than sample size" */
  double* grads = (double*)malloc(sizeof(double) * np);
  double* Xb = (double*)malloc(sizeof(double) * (size_t)batch * F);
  double* Yb = (double*)malloc(sizeof(double) * (size_t)batch * no);
  double *acts[8] = {0}, *deltas[8] = {0};
  for (int i = 1; i < L; i++) {
    acts[i] = (double*)calloc((size_t)batch * cfg->units[i], sizeof(double));
    deltas[i - 1] = (double*)calloc((size_t)batch * cfg->units[i], sizeof(double));
  }
  double best = INFINITY;
  int no_improve = 0, it = 0;
  for (it = 0; it < max_iter; it++) {
    double acc = 0;
    for (int64_t s = 0; s < n; s += batch) {                      /* :790 */
      const int ns = (int)(s + batch > n ? n - s : batch);       /* :791-793 */
      const double *xb = X + s * F, *yb = Y + s * no;
      if (perm) {
        const int32_t* p = perm + (int64_t)it * n + s;
        for (int r = 0; r < ns; r++) {
          memcpy(Xb + (size_t)r * F, X + (int64_t)p[r] * F, sizeof(double) * (size_t)F);
          memcpy(Yb + (size_t)r * no, Y + (int64_t)p[r] * no, sizeof(double) * (size_t)no);
        }
        xb = Xb; yb = Yb;
      }
      double bl = orc_mlp_loss_grad_rows(cfg, theta, xb, yb, ns, batch, acts, deltas, grads);
      acc += bl * (double)ns;                                     /* :806 */
      orc_mlp_update(opt, theta, grads, np);
    }
    double loss = acc / (double)n;                                /* :812 */
    loss_curve[it] = loss;
    /* updateNoImprovementCount :859-895 (no early stopping) */
    if (loss > best - tol) no_improve++; else no_improve = 0;
    if (loss < best) best = loss;
    if (no_improve > n_iter_no_change) { it++; break; } /* constant schedule => stop :826-835 */
  }
  for (int i = 1; i < L; i++) { free(acts[i]); free(deltas[i - 1]); }
  free(grads); free(Xb); free(Yb);
  return it;
}
