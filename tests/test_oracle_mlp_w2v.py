"""Oracle checks for the sklearn-port MLP (nn/neural_network/basemlp64.go) and item2vec
(feature/embedding/model/word2vec) restatements: independent numpy / pure-Python restatements of the
same reference lines, finite differences, scikit-learn's own loss, and fast-vs-literal Huffman."""
import math

import numpy as np
import pytest


# ------------------------------------------------------------------ MLP --
def np_mlp_loss_grad(units, theta, X, Y, act, alpha):
    """numpy float64 restatement of backprop (basemlp64.go:340-406)."""
    off, Ws, bs = 0, [], []
    for i in range(len(units) - 1):
        bs.append(theta[off:off + units[i + 1]]); off += units[i + 1]
        Ws.append(theta[off:off + units[i] * units[i + 1]].reshape(units[i], units[i + 1])); off += units[i] * units[i + 1]
    acts = [X]
    for i in range(len(Ws)):
        z = acts[-1] @ Ws[i] + bs[i]
        if i + 1 != len(Ws):
            z = {"relu": lambda v: np.maximum(v, 0), "logistic": lambda v: 1 / (1 + np.exp(-v)),
                 "tanh": lambda v: np.tanh(-v), "identity": lambda v: v}[act](z)
        else:
            z = 1 / (1 + np.exp(-z))
        acts.append(z)
    n = X.shape[0]
    h = np.clip(acts[-1], np.nextafter(0, 1), np.nextafter(1, 0))
    loss = (-(Y * np.log(h)) - (1 - Y) * np.log1p(-h)).sum() / n + 0.5 * alpha * sum((W * W).sum() for W in Ws) / n
    grads = np.zeros_like(theta)
    delta = acts[-1] - Y
    gW, gb = [None] * len(Ws), [None] * len(Ws)
    for i in range(len(Ws) - 1, -1, -1):
        gW[i] = acts[i].T @ delta / n + alpha / n * Ws[i]
        gb[i] = delta.mean(0)
        if i >= 1:
            delta = delta @ Ws[i].T
            a = acts[i]
            if act == "relu":
                delta = np.where(a == 0, 0.0, delta)
            elif act == "logistic":
                delta = delta * a * (1 - a)
            elif act == "tanh":
                delta = delta * (1 - a * a)
    off = 0
    for i in range(len(Ws)):
        grads[off:off + units[i + 1]] = gb[i]; off += units[i + 1]
        grads[off:off + Ws[i].size] = gW[i].ravel(); off += Ws[i].size
    return loss, grads, acts[-1]


@pytest.mark.parametrize("act", ["relu", "logistic", "tanh", "identity"])
@pytest.mark.parametrize("units", [[6, 4, 1], [281, 100, 1], [9, 7, 5, 1]])
def test_mlp_loss_grad_vs_numpy(oracle, act, units):
    rng = np.random.default_rng(1)
    cfg = oracle.mlp_cfg(units, act, alpha=1e-2)
    n = oracle.mlp_nparams(cfg)
    assert n == sum((1 + units[i]) * units[i + 1] for i in range(len(units) - 1))
    theta = rng.random(n) * 0.3  # one-sided init like Q8
    X = rng.random((40, units[0])); Y = (rng.random((40, 1)) < 0.5).astype(np.float64)
    loss, g = oracle.mlp_loss_grad(cfg, theta, X, Y)
    rloss, rg, rout = np_mlp_loss_grad(units, theta, X, Y, act, 1e-2)
    assert loss == pytest.approx(rloss, rel=1e-12)
    assert np.allclose(g, rg, rtol=1e-10, atol=1e-14)
    assert np.allclose(oracle.mlp_predict(cfg, theta, X), rout, rtol=1e-12, atol=0)


def test_mlp_gradient_finite_difference(oracle):
    # multilayer_perceptron_test.go:118-130 does the same check on the reference
    rng = np.random.default_rng(2)
    units = [6, 4, 1]
    cfg = oracle.mlp_cfg(units, "logistic", alpha=1.0)
    theta = rng.random(oracle.mlp_nparams(cfg)) - 0.5
    X = rng.random((30, 6)); Y = (rng.random((30, 1)) < 0.5).astype(np.float64)
    _, g = oracle.mlp_loss_grad(cfg, theta.copy(), X, Y)
    eps = 1e-6
    for i in range(theta.size):
        tp, tm = theta.copy(), theta.copy()
        tp[i] += eps; tm[i] -= eps
        fd = (oracle.mlp_loss_grad(cfg, tp, X, Y)[0] - oracle.mlp_loss_grad(cfg, tm, X, Y)[0]) / (2 * eps)
        assert fd == pytest.approx(g[i], abs=1e-7)


def test_mlp_loss_equals_sklearn_log_loss(oracle):
    from sklearn.metrics import log_loss
    rng = np.random.default_rng(3)
    cfg = oracle.mlp_cfg([5, 3, 1], "relu", alpha=0.0)
    theta = rng.random(oracle.mlp_nparams(cfg))
    X = rng.random((25, 5)); Y = (rng.random((25, 1)) < 0.5).astype(np.float64)
    loss, _ = oracle.mlp_loss_grad(cfg, theta.copy(), X, Y)
    p = oracle.mlp_predict(cfg, theta, X)
    assert loss == pytest.approx(log_loss(Y.ravel(), p.ravel(), labels=[0, 1]), rel=1e-12)


def test_mlp_adam_per_parameter_beta_powers(oracle):
    """Q7 (basemlp64.go:1082-1087): exponent for parameter i at step s is (s-1)*n + i + 1."""
    n = 7
    rng = np.random.default_rng(4)
    theta = rng.random(n); ref = theta.copy()
    opt = oracle.MlpOptimizer("adam", n, lr_init=0.001)
    m = np.zeros(n); v = np.zeros(n)
    for s in range(1, 4):
        g = rng.standard_normal(n)
        opt.update(theta, g)
        for i in range(n):
            m[i] = 0.9 * m[i] + 0.1 * g[i]
            v[i] = 0.999 * v[i] + 0.001 * g[i] * g[i]
            e = (s - 1) * n + i + 1
            lr = 0.001 * math.sqrt(1 - 0.999 ** e) / (1 - 0.9 ** e)
            ref[i] += -lr * m[i] / (math.sqrt(v[i]) + 1e-8)
    assert np.allclose(theta, ref, rtol=1e-9, atol=1e-15)


def test_mlp_batchnorm_maxabs(oracle):
    """Q10 (basemlp64.go:277-308): hidden activations scaled by column max-abs AFTER the forward pass;
    deltas divided by M."""
    rng = np.random.default_rng(5)
    units = [4, 3, 1]
    cfg = oracle.mlp_cfg(units, "relu", alpha=0.0, batch_normalize=True)
    cfg0 = oracle.mlp_cfg(units, "relu", alpha=0.0)
    theta = rng.random(oracle.mlp_nparams(cfg))
    X = rng.random((10, 4)); Y = (rng.random((10, 1)) < 0.5).astype(np.float64)
    loss_bn, g_bn = oracle.mlp_loss_grad(cfg, theta.copy(), X, Y)
    loss0, g0 = oracle.mlp_loss_grad(cfg0, theta.copy(), X, Y)
    assert loss_bn == loss0                       # the output layer saw un-normalised activations
    b1, W1 = theta[:3], theta[3:15].reshape(4, 3)
    a1 = np.maximum(X @ W1 + b1, 0); M = np.abs(a1).max(0)
    # last-layer coef grads use normalised a1
    h = oracle.mlp_predict(cfg0, theta, X)
    gW2 = (a1 / M).T @ (h - Y) / 10
    assert np.allclose(g_bn[15 + 1:15 + 1 + 3], gW2.ravel(), rtol=1e-12)
    assert not np.allclose(g_bn, g0)


def test_mlp_fit_loss_decreases_and_stops(oracle):
    rng = np.random.default_rng(6)
    units = [8, 6, 1]
    cfg = oracle.mlp_cfg(units, "relu", alpha=1e-5)
    n = oracle.mlp_nparams(cfg)
    theta = rng.random(n) * math.sqrt(6 / (8 + 6))
    X = rng.random((400, 8)); Y = ((X[:, :1] + X[:, 1:2]) > 1).astype(np.float64)
    opt = oracle.MlpOptimizer("adam", n)
    curve = oracle.mlp_fit(cfg, theta, opt, X, Y, batch=200, max_iter=40)
    assert curve[-1] < curve[0]
    assert len(curve) <= 40


# -------------------------------------------------------------- item2vec --
@pytest.mark.parametrize("seed", range(6))
def test_huffman_fast_equals_literal(oracle, seed):
    rng = np.random.default_rng(seed)
    V = int(rng.integers(1, 200))
    counts = rng.integers(1, 6, size=V)  # many ties -> exercises huffman.go:44-46 insertion order
    a = oracle.huffman_paths(counts, max_depth=100)
    b = oracle.huffman_paths(counts, max_depth=100, slow=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_huffman_properties_and_depth_truncation(oracle):
    counts = np.array([5, 9, 12, 13, 16, 45])
    off, nodes, codes = oracle.huffman_paths(counts)
    lens = np.diff(off)
    assert lens.tolist() == [4, 4, 3, 3, 3, 1]         # classic example: code lengths
    assert 2.0 ** (-lens.astype(float)).sum() or True
    assert sum(2.0 ** -l for l in lens) == 1.0          # Kraft equality for a full tree
    # every path starts at the root = last-created inner node (V-2)
    assert all(nodes[off[i]] == len(counts) - 2 for i in range(len(counts)))
    # node.go:39-42: path keeps at most max_depth nodes incl. leaf => max_depth-1 pairs
    off2, _, _ = oracle.huffman_paths(counts, max_depth=3)
    assert np.diff(off2).tolist() == [2, 2, 2, 2, 2, 1]
    assert oracle.huffman_paths(np.array([7]))[0].tolist() == [0, 0]   # V == 1: no inner node


def py_w2v_pass(cfg, doc, keep, param, node, paths, sigtab, lr, corpus_len, update_lr_batch, min_lr, init_lr):
    """Pure-Python restatement of word2vec.go:198-243 + model.go:48-78 + optimizer.go:107-129."""
    off, nodes, codes = paths
    nxt = 1
    cnt = 0
    dim, win = param.shape[1], cfg.window
    for pos, wid in enumerate(doc):
        if keep is None or keep[pos]:
            nxt = (nxt * 25214903917 + 11) % (1 << 64)
            d = nxt % win
            for a in range(d, win * 2 + 1 - d):
                if a == win:
                    continue
                c = pos - win + a
                if c < 0 or c >= len(doc):
                    continue
                tmp = [0.0] * dim
                ctx = param[doc[c]]
                for i in range(off[wid], off[wid + 1]):
                    pv = node[nodes[i]]
                    inner = 0.0
                    for j in range(dim):
                        inner += ctx[j] * pv[j]
                    if inner <= -6.0 or inner >= 6.0:
                        break
                    g = (1.0 - float(codes[i]) - sigtab[int((inner + 6.0) * (1000 / 6.0 / 2.0))]) * lr
                    for j in range(dim):
                        tmp[j] += g * pv[j]
                        pv[j] += g * ctx[j]
                for j in range(dim):
                    ctx[j] += tmp[j]
        cnt += 1
        if cnt % update_lr_batch == 0:
            lr = min_lr if lr < min_lr else init_lr * (1.0 - cnt / corpus_len)
    return lr


def test_w2v_single_stream_matches_python_restatement(oracle):
    rng = np.random.default_rng(7)
    V, dim, n = 12, 4, 400
    doc = rng.integers(0, V, size=n).astype(np.int32)
    counts = np.bincount(doc, minlength=V)
    paths = oracle.huffman_paths(counts)
    sig = oracle.sigmoid_table()
    keep = (rng.random(n) < 0.9).astype(np.uint8)
    param0 = (rng.random((V, dim)) - 0.5) / dim
    cfg = oracle.w2v_cfg(dim=dim, window=5, update_lr_batch=100)
    p1, n1 = param0.copy(), np.zeros((V - 1, dim))
    lcg = oracle.Lcg(1)
    lr1, cnt = oracle.w2v_train_slice(cfg, doc, 0, n, keep, p1, n1, paths, sig, lcg, 0.025, 0, n + 50)
    p2, n2 = param0.copy(), np.zeros((V - 1, dim))
    lr2 = py_w2v_pass(cfg, doc.tolist(), keep, p2, n2, paths, sig, 0.025, n + 50, 100, cfg.min_lr, 0.025)
    assert cnt == n and lr1 == lr2
    assert np.array_equal(p1, p2) and np.array_equal(n1, n2)   # bit-exact float64
    assert not np.array_equal(p1, param0)


def test_w2v_slice_window_is_clipped_to_slice(oracle):
    """Q18: window clipping is against the thread's slice, not the whole doc."""
    rng = np.random.default_rng(8)
    V, dim, n = 10, 4, 120
    doc = rng.integers(0, V, size=n).astype(np.int32)
    paths = oracle.huffman_paths(np.bincount(doc, minlength=V))
    sig = oracle.sigmoid_table()
    cfg = oracle.w2v_cfg(dim=dim)
    pa, na = (rng.random((V, dim)) - 0.5) / dim, np.zeros((V - 1, dim))
    pb, nb = pa.copy(), na.copy()
    oracle.w2v_train_slice(cfg, doc, 40, 80, None, pa, na, paths, sig, oracle.Lcg(1), 0.025, 0, n)
    oracle.w2v_train_slice(cfg, doc[40:80].copy(), 0, 40, None, pb, nb, paths, sig, oracle.Lcg(1), 0.025, 0, n)
    assert np.array_equal(pa, pb) and np.array_equal(na, nb)


@pytest.mark.parametrize("opt,model", [("ns", "skipgram"), ("hs", "cbow"), ("ns", "cbow")])
def test_w2v_other_modes_run_and_are_deterministic(oracle, opt, model):
    rng = np.random.default_rng(9)
    V, dim, n = 15, 4, 300
    doc = rng.integers(0, V, size=n).astype(np.int32)
    paths = oracle.huffman_paths(np.bincount(doc, minlength=V))
    sig = oracle.sigmoid_table()
    cfg = oracle.w2v_cfg(dim=dim, optimizer=opt, model=model)
    res = []
    for _ in range(2):
        p = (np.random.default_rng(1).random((V, dim)) - 0.5) / dim
        aux = np.zeros((V - 1, dim)) if opt == "hs" else (np.random.default_rng(2).random((V, dim)) - 0.5) / dim
        oracle.w2v_train_slice(cfg, doc, 0, n, None, p, aux, paths, sig, oracle.Lcg(1), 0.025, 0, n)
        res.append((p, aux))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.all(np.isfinite(res[0][0]))
