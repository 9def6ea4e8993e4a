"""Quirk Q11 -- the SHORT LAST BATCH of the sklearn-port MLP (nn/neural_network/basemlp64.go:790-812).

The reference's own entry point trains one: main.go:39-50 fits 79 948 rows at the default batch of 200 = 399 full batches
and one of 148.  What that batch computes follows from three facts of the Go source:

  * fit allocates ONE activation and ONE delta block of BatchSize rows per layer for the whole call (:529-545);
  * `for _, a := range activations { a.Rows = Xbatch.Rows }` (:800-802) assigns to a COPY of each header: only
    activations[0] = Xbatch (:799) has the short row count, the other headers keep Rows = BatchSize;
  * gonum's blas64.Gemm(tA, tB, alpha, A, B, beta, C) takes m, k from A's header and n from B's, and never looks at
    C.Rows (gonum v0.11.0 blas64/blas64.go); every element-wise helper loops over ITS argument's Rows.

This file does NOT re-derive the consequences by hand: it emulates the headers (`Gen`: Rows, Cols, a view of the backing
array) and gonum's Gemm rule, and transcribes forwardPass / backprop / computeLossGrad call by call.  The oracle's C
restatement (oracle/orc_sklmlp.c: orc_mlp_loss_grad_rows, orc_mlp_fit) must agree with it -- the oracle is what the device
is then compared with (tests/test_gpu_mlp.py)."""
import numpy as np
import pytest

from oracle import pyoracle


class Gen:
    """blas64.General: a header over a backing array"""
    def __init__(self, rows, cols, data):
        self.Rows, self.Cols, self.Data = rows, cols, data      # Data: 2-D numpy view [>= rows, cols] (Stride = Cols)

    def m(self):
        return self.Data[:self.Rows]


def gemm(tA, tB, alpha, a, b, beta, c):
    """gonum blas64.Gemm: m, k from a's header; n from b's; Dgemm then addresses b as a k x n (or n x k) block of its
    backing array and c as m x n -- neither b's other dimension nor c's Rows is ever read"""
    A = a.m().T if tA else a.m()
    m, k = A.shape
    n = b.Rows if tB else b.Cols
    Bm = b.Data[:n, :k].T if tB else b.Data[:k, :n]
    assert beta == 0
    c.Data[:m, :n] = alpha * (A @ Bm)


def add_intercepts(a, b):                                        # :205-211, a.Rows rows
    a.Data[:a.Rows] += b


def relu(z):                                                      # :98-106, z.Rows rows
    v = z.Data[:z.Rows]
    v[v < 0] = 0


def logistic(z):
    z.Data[:z.Rows] = 1 / (1 + np.exp(-z.Data[:z.Rows]))


def binary_log_loss(y, h):                                        # :180-195: rows of y, divided by h.Rows
    hv = np.clip(h.Data[:y.Rows], np.nextafter(0.0, 1.0), np.nextafter(1.0, 0.0))
    yv = y.m()
    return float(np.sum(-yv * np.log(hv) - (1 - yv) * np.log1p(-hv))) / h.Rows


class GoMlp:
    """[F, H, 1] relu / logistic network with the reference's buffers"""
    def __init__(self, F, H, B, alpha, rng):
        self.W = [rng.standard_normal((F, H)) * 0.3, rng.standard_normal((H, 1)) * 0.3]
        self.b = [rng.standard_normal(H) * 0.1, rng.standard_normal(1) * 0.1]
        self.alpha, self.B = alpha, B
        mem = [np.zeros((B, H)), np.zeros((B, 1))]
        self.activations = [None, Gen(B, H, mem[0]), Gen(B, 1, mem[1])]         # fit :529-545
        self.deltas = [Gen(B, H, np.zeros((B, H))), Gen(B, 1, np.zeros((B, 1)))]

    def theta(self):
        return np.concatenate([self.b[0], self.W[0].ravel(), self.b[1], self.W[1].ravel()])

    def backprop(self, Xb, Yb):
        """fitStochastic :795-805 + backprop :340-406, call by call"""
        act, dl = self.activations, self.deltas
        X = Gen(Xb.shape[0], Xb.shape[1], Xb)
        y = Gen(Yb.shape[0], 1, Yb)
        act[0] = X
        for a in act:                       # :800-802 -- `a` is a copy of the header: no effect
            a_copy = Gen(a.Rows, a.Cols, a.Data)
            a_copy.Rows = X.Rows
        nSamples = X.Rows
        Wg = [Gen(*w.shape, w) for w in self.W]
        # forwardPass :259-274
        for i in range(2):
            gemm(False, False, 1, act[i], Wg[i], 0, act[i + 1])
            add_intercepts(act[i + 1], self.b[i])
            if i + 1 != 2:
                relu(act[i + 1])
        logistic(act[2])
        loss = binary_log_loss(y, act[2])
        loss += 0.5 * self.alpha * sum(float(np.sum(w * w)) for w in self.W) / nSamples
        H, D = act[2], dl[1]
        D.Data[:y.Rows] = H.Data[:y.Rows] - y.m()                # :373-381 (y.Rows rows)
        cg = [Gen(*w.shape, np.zeros_like(w)) for w in self.W]
        ig = [np.zeros_like(b) for b in self.b]

        def compute_loss_grad(layer):                            # :322-330
            gemm(True, False, 1 / nSamples, act[layer], dl[layer], 0, cg[layer])
            cg[layer].Data += self.alpha / nSamples * self.W[layer]
            ig[layer][:] = dl[layer].m().sum(0) / dl[layer].Rows  # matRowMean64 :213-226

        compute_loss_grad(1)
        gemm(False, True, 1, dl[1], Wg[1], 0, dl[0])             # :388
        z, d = act[1], dl[0]
        d.Data[:z.Rows][z.Data[:z.Rows] == 0] = 0                # relu' :139-147 (Z.Rows rows)
        compute_loss_grad(0)
        return loss, np.concatenate([ig[0], cg[0].Data.ravel(), ig[1], cg[1].Data.ravel()])


@pytest.mark.parametrize("ns", [200, 148, 1])
def test_oracle_short_last_batch_is_the_go_semantics(ns):
    F, Hh, B, alpha = 9, 7, 200, 1e-3
    rng = np.random.default_rng(5)
    g = GoMlp(F, Hh, B, alpha, rng)
    cfg = pyoracle.mlp_cfg([F, Hh, 1], "relu", alpha=alpha)
    acts, deltas = pyoracle.mlp_blocks(cfg, B)
    theta = g.theta()
    # a full batch first: it leaves the rows the short one will see
    X0, Y0 = rng.random((B, F)), (rng.random((B, 1)) < 0.5).astype(np.float64)
    l0, g0 = g.backprop(X0.copy(), Y0.copy())
    o0, og0 = pyoracle.mlp_loss_grad_rows(cfg, theta, X0, Y0, B, acts, deltas)
    assert abs(l0 - o0) <= 1e-13 and np.max(np.abs(g0 - og0)) <= 1e-13
    X1, Y1 = rng.random((ns, F)), (rng.random((ns, 1)) < 0.5).astype(np.float64)
    l1, g1 = g.backprop(X1.copy(), Y1.copy())
    o1, og1 = pyoracle.mlp_loss_grad_rows(cfg, theta, X1, Y1, B, acts, deltas)
    assert abs(l1 - o1) <= 1e-13 and np.max(np.abs(g1 - og1)) <= 1e-13
    np.testing.assert_allclose(acts[1], g.activations[1].Data, atol=1e-13)
    np.testing.assert_allclose(deltas[1], g.deltas[1].Data, atol=1e-13)
    if ns < B:
        # what the quirk IS: the short batch's gradient is not the plain gradient of its ns rows ...
        pl, pg = pyoracle.mlp_loss_grad(cfg, theta, X1, Y1)
        assert np.max(np.abs(pg - og1)) > 1e-6 and abs(pl - o1) > 1e-6
        # ... except for the first layer's coefficient block, which only ever sees the ns fresh rows
        w0 = slice(Hh, Hh + F * Hh)
        np.testing.assert_allclose(og1[w0], pg[w0], atol=1e-13)


def test_oracle_fit_trains_the_short_batch_and_divides_by_all_rows():
    """fitStochastic :790-812: ceil(n / B) updates per epoch, loss = sum(batch loss x batch rows) / nSamples"""
    F, Hh, B, n = 6, 5, 16, 16 * 3 + 5
    rng = np.random.default_rng(2)
    cfg = pyoracle.mlp_cfg([F, Hh, 1], "relu", alpha=1e-4)
    X, Y = rng.random((n, F)), (rng.random((n, 1)) < 0.5).astype(np.float64)
    th0 = rng.standard_normal(pyoracle.mlp_nparams(cfg)) * 0.3
    perm = np.stack([rng.permutation(n) for _ in range(3)]).astype(np.int32)
    th = th0.copy()
    opt = pyoracle.MlpOptimizer("adam", th.size)
    curve = pyoracle.mlp_fit(cfg, th, opt, X, Y, B, 3, perm=perm)
    # the same through single calls
    th2 = th0.copy()
    opt2 = pyoracle.MlpOptimizer("adam", th2.size)
    acts, deltas = pyoracle.mlp_blocks(cfg, B)
    want = []
    for it in range(3):
        acc = 0.0
        for s in range(0, n, B):
            idx = perm[it, s:s + B]
            l, g = pyoracle.mlp_loss_grad_rows(cfg, th2, X[idx], Y[idx], B, acts, deltas)
            acc += l * len(idx)
            opt2.update(th2, g)
        want.append(acc / n)
    assert opt.o.t == 3 * 4                                      # four updates per epoch, the short batch included
    np.testing.assert_array_equal(curve, np.array(want))
    np.testing.assert_array_equal(th, th2)
