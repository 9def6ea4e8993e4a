"""CPU: the oracle of the trainable-embedding EXTENSION (oracle/orc_embtrain.c).  The reference freezes the embeddings
(SURVEY F3), so nothing of go-ctr pins this: the float64 forward must agree with the float32 restatement of the
reference's graph (orc_ctr.c, itself pinned on the op-level KATs), and the analytic gradient with central finite
differences of the same loss."""
import numpy as np
import pytest


def _case(oracle, kind, att, seed=0, B=6, U=5, T=7, D=8, Cc=4, V=23, H1=16, H2=8, scale=0.4):
    rng = np.random.default_rng(seed)
    m = oracle.CtrModel(kind, U, T, D, Cc, H1=H1, H2=H2, att=att)
    m.W0[:] = rng.standard_normal(m.W0.shape) * scale
    m.W1[:] = rng.standard_normal(m.W1.shape) * scale
    m.W2[:] = rng.standard_normal(m.W2.shape) * scale
    m.att0[:] = 1.0 + 0.3 * rng.standard_normal(T)
    E = rng.standard_normal((V, D)) * 0.5
    ub = rng.integers(-2, V + 1, size=(B, T)).astype(np.int32)       # some empty (-1, -2) and out-of-range (V) slots
    ub[0, :3] = ub[0, 3]                                             # one id several times in a history
    items = rng.integers(0, V, size=B).astype(np.int32)
    ub[1, 0] = items[1]                                              # history contains the candidate itself
    items[2] = -1                                                    # unknown candidate
    uf = rng.random((B, U), dtype=np.float32)
    cf = rng.random((B, Cc), dtype=np.float32)
    y = (rng.random(B) < 0.5).astype(np.float32)
    return m, E, ub, items, uf, cf, y


@pytest.mark.parametrize("kind,att", [("youtube", 0), ("din", 0), ("din", 1)])
def test_f64_forward_matches_f32_restatement(oracle, kind, att):
    k = oracle.DIN if kind == "din" else oracle.YOUTUBE
    m, E, ub, items, uf, cf, y = _case(oracle, k, att)
    E32 = E.astype(np.float32)
    loss64 = m.emb_loss_grad(E32.astype(np.float64), ub, items, uf, cf, y, B=8, want_grad=False)   # 2 padded rows
    X = oracle.assemble_rows(E32, ub, items, uf, cf)
    loss32, _, _ = m.loss_grad(X, y, B=8)
    assert abs(loss64 - loss32) < 2e-6 * max(1.0, abs(loss64))


@pytest.mark.parametrize("kind,att", [("youtube", 0), ("din", 0), ("din", 1)])
def test_gradient_vs_finite_differences(oracle, kind, att):
    k = oracle.DIN if kind == "din" else oracle.YOUTUBE
    m, E, ub, items, uf, cf, y = _case(oracle, k, att, seed=3)
    drop = {"mode": 2, "p0": 0.2, "p1": 0.2, "seed": 5, "step": 3}     # hash masks: constant under perturbation
    loss, dE = m.emb_loss_grad(E, ub, items, uf, cf, y, B=8, drop=drop)
    used = set(ub[(ub >= 0) & (ub < E.shape[0])].tolist()) | set(items[items >= 0].tolist())
    assert np.all(dE[[i for i in range(E.shape[0]) if i not in used]] == 0.0)
    rng = np.random.default_rng(1)
    eps = 1e-6
    checked = 0
    for i in sorted(used):
        for d in rng.choice(E.shape[1], size=2, replace=False):
            Ep, Em = E.copy(), E.copy()
            Ep[i, d] += eps
            Em[i, d] -= eps
            fd = (m.emb_loss_grad(Ep, ub, items, uf, cf, y, B=8, drop=drop, want_grad=False) -
                  m.emb_loss_grad(Em, ub, items, uf, cf, y, B=8, drop=drop, want_grad=False)) / (2 * eps)
            assert abs(fd - dE[i, d]) < 1e-7 + 1e-5 * abs(fd), (i, d, fd, dE[i, d])
            checked += 1
    assert checked > 20 and np.abs(dE).max() > 1e-4
