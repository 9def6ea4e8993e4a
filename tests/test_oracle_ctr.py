"""Cross-check the oracle's hand-derived DIN / YouTube forward+backward against torch autograd
(an independent implementation of the same graph, float64 on CPU), the Adam step against a numpy
restatement, and the train/predict loops' padding semantics (model.go:132-136,357-371)."""
import numpy as np
import pytest
import torch


def make_data(rng, rows, U, T, D, Cc, pad_frac=0.3):
    X = rng.random((rows, U + T * D + D + Cc), dtype=np.float32)
    # zero some behaviour slots like rcmd.go:517-522 padding
    ub = X[:, U:U + T * D].reshape(rows, T, D)
    ub[rng.random((rows, T)) < pad_frac] = 0.0
    Y = (rng.random(rows) < 0.5).astype(np.float32)
    return X, Y


def torch_model(m, X, Y, B, kind, att, drop=None):
    """Same graph as din.go:219-323 / dnn.go:162-184 + cost.go:9-17 in float64 torch."""
    cfg = m.cfg
    U, T, D, Cc = cfg.U, cfg.T, cfg.D, cfg.C
    Xp = np.zeros((B, X.shape[1]), np.float64)
    Xp[:X.shape[0]] = X
    Yp = np.zeros(B, np.float64)
    Yp[:len(Y)] = Y
    x = torch.tensor(Xp)
    y = torch.tensor(Yp)
    W0 = torch.tensor(m.W0.astype(np.float64), requires_grad=True)
    W1 = torch.tensor(m.W1.astype(np.float64), requires_grad=True)
    W2 = torch.tensor(m.W2.astype(np.float64), requires_grad=True)
    a0 = torch.tensor(m.att0.astype(np.float64), requires_grad=True)
    u = x[:, :U]
    ub = x[:, U:U + T * D].reshape(B, T, D)
    v = x[:, U + T * D:U + T * D + D]
    c = x[:, U + T * D + D:]
    if kind == 0:
        if att == 0:
            s = (ub * v[:, None, :]).sum(-1)
            cos = s / (ub.pow(2).sum(-1).sqrt() * v.pow(2).sum(-1).sqrt()[:, None] + 1e-8)
            w = (cos + 1) / 2
        else:
            w = 1 - (ub - v[:, None, :]).pow(2).sum(-1).sqrt()
        g = torch.sigmoid(w * a0[None, :])
        p = (g[:, :, None] * ub).mean(1)
    else:
        p = ub.mean(1)
    h0 = torch.cat([u, p, v, c], 1)
    A0 = torch.sigmoid(h0 @ W0)
    if drop is not None:
        A0 = A0 * torch.tensor(drop["m0"].astype(np.float64)) / (1 - drop["p0"])
    A1 = torch.sigmoid(A0 @ W1)
    if drop is not None:
        A1 = A1 * torch.tensor(drop["m1"].astype(np.float64)) / (1 - drop["p1"])
    yh = torch.sigmoid(A1 @ W2)[:, 0]
    cost = -(y * yh.log() + (1 - y) * (1 - yh).log()).mean()
    cost.backward()
    grads = dict(W0=W0.grad.numpy(), W1=W1.grad.numpy(), W2=W2.grad.numpy(),
                 att0=a0.grad.numpy() if a0.grad is not None else np.zeros(T))
    return yh.detach().numpy(), float(cost), grads


def small_weights(m, rng, scale):
    m.W0[:] = (rng.standard_normal(m.W0.shape) * scale).astype(np.float32)
    m.W1[:] = (rng.standard_normal(m.W1.shape) * scale).astype(np.float32)
    m.W2[:] = (rng.standard_normal(m.W2.shape) * scale).astype(np.float32)
    m.att0[:] = (1 + 0.3 * rng.standard_normal(m.att0.shape)).astype(np.float32)


@pytest.mark.parametrize("kind,att", [(0, 0), (0, 1), (1, 0)])
@pytest.mark.parametrize("dims", [(5, 3, 7, 5, 8, 8), (52, 10, 16, 53, 200, 64)])
def test_forward_backward_vs_torch(oracle, kind, att, dims):
    U, T, D, Cc, B, valid = dims
    rng = np.random.default_rng(10 * kind + att)
    m = oracle.CtrModel(kind, U, T, D, Cc, att=att)
    small_weights(m, rng, 0.15)
    X, Y = make_data(rng, valid, U, T, D, Cc)
    cost, g, y = m.loss_grad(X, Y, B=B)
    ty, tcost, tg = torch_model(m, X, Y, B, kind, att)
    assert np.max(np.abs(y - ty)) < 2e-6          # logits, fp32 vs fp64 graph
    assert abs(cost - tcost) < 2e-6
    for k in ("W0", "W1", "W2") + (("att0",) if kind == 0 else ()):
        ref = tg[k].reshape(g[k].shape)
        err = np.max(np.abs(g[k] - ref))
        assert err <= 1e-6 + 2e-4 * np.max(np.abs(ref)), (k, err)


def test_padded_batch_counts_in_loss_and_grads(oracle):
    # model.go:132-136: short batch is zero-padded to B and the pad rows take part (quirk Q3)
    U, T, D, Cc = 5, 3, 7, 5
    rng = np.random.default_rng(3)
    m = oracle.CtrModel(0, U, T, D, Cc)
    small_weights(m, rng, 0.2)
    X, Y = make_data(rng, 18, U, T, D, Cc)
    cost_pad, g_pad, _ = m.loss_grad(X, Y, B=20)
    Xz = np.vstack([X, np.zeros((2, X.shape[1]), np.float32)])
    Yz = np.concatenate([Y, np.zeros(2, np.float32)])
    cost_full, g_full, _ = m.loss_grad(Xz, Yz, B=20)
    assert cost_pad == cost_full
    for k in g_pad:
        assert np.array_equal(g_pad[k], g_full[k])
    ty, tcost, tg = torch_model(m, X, Y, 20, 0, 0)
    assert abs(cost_pad - tcost) < 2e-6


def test_injected_dropout_mask_vs_torch(oracle):
    U, T, D, Cc, B = 5, 3, 7, 5, 16
    rng = np.random.default_rng(5)
    m = oracle.CtrModel(0, U, T, D, Cc)
    small_weights(m, rng, 0.2)
    X, Y = make_data(rng, B, U, T, D, Cc)
    drop = dict(mode=1, p0=0.25, p1=0.5, m0=(rng.random((B, 200)) < 0.75).astype(np.float32),
                m1=(rng.random((B, 80)) < 0.5).astype(np.float32))
    cost, g, y = m.loss_grad(X, Y, drop=drop)
    ty, tcost, tg = torch_model(m, X, Y, B, 0, 0, drop=drop)
    assert np.max(np.abs(y - ty)) < 2e-6 and abs(cost - tcost) < 2e-6
    for k in g:
        ref = tg[k].reshape(g[k].shape)
        assert np.max(np.abs(g[k] - ref)) <= 1e-6 + 2e-4 * np.max(np.abs(ref))


def test_hash_dropout_is_deterministic_and_has_right_rate(oracle):
    import ctypes as C
    L = oracle.lib()
    keep = np.array([[L.orc_dropout_keep(7, 3, 0, r, c, C.c_float(0.25)) for c in range(200)] for r in range(64)])
    assert set(np.unique(keep)) <= {0.0, 1.0}
    assert 0.70 < keep.mean() < 0.80
    keep2 = np.array([[L.orc_dropout_keep(7, 3, 0, r, c, C.c_float(0.25)) for c in range(200)] for r in range(64)])
    assert np.array_equal(keep, keep2)
    keep3 = np.array([[L.orc_dropout_keep(7, 4, 0, r, c, C.c_float(0.25)) for c in range(200)] for r in range(64)])
    assert not np.array_equal(keep, keep3)


def numpy_adam(w, g, m, v, it, lr=0.01, l2=1e-4, b1=0.9, b2=0.999, eps=1e-8, batch=1):
    """gorgonia AdamSolver.Step in float32 numpy."""
    f = np.float32
    g = g + w * f(l2)
    if batch > 1:
        g = g * (f(1) / f(batch))
    m = f(b1) * m + (f(1) - f(b1)) * g
    v = f(b2) * v + (g * g) * (f(1) - f(b2))
    c1 = f(1) / f(1 - b1 ** it)
    c2 = f(1) / f(1 - b2 ** it)
    w = w + (f(-lr) * (m * c1)) / (np.sqrt(v * c2) + f(eps))
    return w.astype(f), m.astype(f), v.astype(f)


def test_train_loop_matches_stepwise_restatement(oracle):
    """orc_ctr_train == loop of loss_grad + numpy Adam, including the padded last batch and
    'cost of the last batch' bookkeeping (model.go:107-211)."""
    U, T, D, Cc = 5, 3, 7, 5
    rng = np.random.default_rng(11)
    m = oracle.CtrModel(0, U, T, D, Cc).init_gaussian(rng)
    m2 = oracle.CtrModel(0, U, T, D, Cc)
    for k in ("W0", "W1", "W2", "att0"):
        getattr(m2, k)[:] = getattr(m, k)
    X, Y = make_data(rng, 50, U, T, D, Cc)
    B, epochs = 16, 3
    costs = m.train(X, Y, batch=B, epochs=epochs)
    st = {k: (np.zeros_like(getattr(m2, k)), np.zeros_like(getattr(m2, k))) for k in ("W0", "W1", "W2", "att0")}
    it, ref_costs = 0, []
    for e in range(epochs):
        for s in range(0, 50, B):
            cost, g, _ = m2.loss_grad(X[s:s + B], Y[s:s + B], B=B)
            it += 1
            for k in st:
                w, mm, vv = numpy_adam(getattr(m2, k), g[k], st[k][0], st[k][1], it, batch=B)
                getattr(m2, k)[:] = w
                st[k] = (mm, vv)
        ref_costs.append(cost)
    assert np.allclose(costs, ref_costs, rtol=0, atol=1e-6)
    for k in st:
        assert np.allclose(getattr(m, k), getattr(m2, k), rtol=0, atol=2e-6), k


def test_early_stop_and_epoch_count(oracle):
    U, T, D, Cc = 5, 3, 7, 5
    rng = np.random.default_rng(12)
    m = oracle.CtrModel(1, U, T, D, Cc).init_gaussian(rng)
    X, Y = make_data(rng, 64, U, T, D, Cc)
    costs = m.train(X, Y, batch=32, epochs=50, early_stop=2)
    # stops as soon as 2 consecutive epochs did not improve on the best cost
    best, noimp = np.inf, 0
    for i, c in enumerate(costs):
        if c < best:
            best, noimp = c, 0
        else:
            noimp += 1
        if noimp >= 2:
            assert i == len(costs) - 1
            break
    else:
        assert len(costs) == 50


def test_predict_padding_118_rows_batch_20(oracle):
    # model_test.go:33-34,104-108: 118 rows at batch 20 exercises the zero-pad path
    U, T, D, Cc = 5, 3, 7, 5
    rng = np.random.default_rng(13)
    m = oracle.CtrModel(0, U, T, D, Cc).init_gaussian(rng)
    X, _ = make_data(rng, 118, U, T, D, Cc)
    y = m.predict(X, batch=20)
    assert y.shape == (118,)
    assert np.array_equal(y, m.forward(X))  # rows are independent in the forward pass
    assert np.all((y >= 0) & (y <= 1))


def test_assemble_rows_layout(oracle):
    # rcmd.go:533: [user | ub(T*D) | itemEmb | itemFeat], missing / negative ids => zero rows
    rng = np.random.default_rng(14)
    V, D, T, U, Cc, rows = 11, 4, 3, 2, 3, 6
    emb = rng.random((V, D), dtype=np.float32)
    ub = rng.integers(-1, V + 2, size=(rows, T)).astype(np.int32)
    it = rng.integers(-1, V + 2, size=rows).astype(np.int32)
    uf = rng.random((rows, U), dtype=np.float32)
    cf = rng.random((rows, Cc), dtype=np.float32)
    X = oracle.assemble_rows(emb, ub, it, uf, cf)
    for r in range(rows):
        exp = list(uf[r])
        for t in range(T):
            exp += list(emb[ub[r, t]]) if 0 <= ub[r, t] < V else [0.0] * D
        exp += list(emb[it[r]]) if 0 <= it[r] < V else [0.0] * D
        exp += list(cf[r])
        assert X[r].tolist() == [float(np.float32(e)) for e in exp]
