"""GPU parity tests for the DIN / YouTube-DNN path: every call goes through the C-ABI
(include/goctr.h via goctr_amd.capi) and is compared with the CPU oracle on the same seeded inputs.
Tolerances: bit-exact for the gather (copies), 1e-5 absolute on logits / loss (BASELINE.json
north_star).  Gradients are bounded against a FLOAT64 evaluation of the same graph: the device may be at
most twice as far from that truth as the float32 oracle itself is (float32 summation order differs: MFMA
k-order and slab sums vs the oracle's sequential loops; the weight-gradient GEMM runs on the 6-product bf16
split).  The dropout cases, which the float64 evaluation does not cover, keep 1e-6 + 2e-4*max|g| vs the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# Multi-epoch training COSTS of the reduced-size shapes below are held to 5e-5, not to the 1e-5 of logits / loss / the full-size
# tests (tests/test_gpu_fullsize.py holds 1e-5 at the benchmark's shapes).  Why: these tests use 0.15-0.3 N(0,1) weights on a few
# dozen inputs, which puts single rows at |z2| ~ 10.  There one float32 ulp of the pre-activation (|z2| 2^-23 ~ 1.2e-6, and device
# and oracle sum z2 in different orders) moves log p = -softplus(-z2) by ~1e-6 for that row; a batch of 20-50 rows holds a few such
# rows, their signs do not cancel inside one mean, and after 2-4 epochs of updates that each saw such a cost the epoch means drift
# by a few 1e-5 (round 3 tightened these from 1e-4 and found 1e-5 too tight for exactly these shapes).  5e-5 = that
# drift with margin; a real defect (a wrong mask, a missing 1/B) shows up at 1e-3 and above.
COST_TOL_SMALL_SHAPES = 5e-5

LOGIT_TOL = 1e-5
LOSS_TOL = 1e-5


def grad_close(g, ref):
    return np.max(np.abs(g - ref)) <= 1e-6 + 2e-4 * np.max(np.abs(ref))


def grad_bound(g_dev, g_orc, g64):
    """|g_dev - truth|_max <= 2 |g_orc32 - truth|_max + one float32 ulp of the largest entry"""
    g_dev, g_orc, g64 = (np.asarray(a, np.float64).ravel() for a in (g_dev, g_orc, g64))
    return np.max(np.abs(g_dev - g64)) <= 2 * np.max(np.abs(g_orc - g64)) + 6e-8 * np.max(np.abs(g64))


def make_data(rng, rows, U, T, D, Cc, pad_frac=0.3):
    X = rng.random((rows, U + T * D + D + Cc), dtype=np.float32)
    ub = X[:, U:U + T * D].reshape(rows, T, D)
    ub[rng.random((rows, T)) < pad_frac] = 0.0
    Y = (rng.random(rows) < 0.5).astype(np.float32)
    return X, Y


def pair(oracle, kind, U, T, D, Cc, rng, att=0, scale=None):
    """an oracle model and a device model with identical weights"""
    from goctr_amd import model as gm
    from goctr_amd.recommend import SampleInfo
    om = oracle.CtrModel(kind, U, T, D, Cc, att=att)
    if scale is None:
        om.init_gaussian(rng)            # the reference's N(0,1) init (din.go:187-191)
    else:
        om.W0[:] = (rng.standard_normal(om.W0.shape) * scale).astype(np.float32)
        om.W1[:] = (rng.standard_normal(om.W1.shape) * scale).astype(np.float32)
        om.W2[:] = (rng.standard_normal(om.W2.shape) * scale).astype(np.float32)
    om.att0[:] = (1 + 0.3 * rng.standard_normal(om.att0.shape)).astype(np.float32) if kind == 0 else 1.0
    cls = gm.DinNet if kind == 0 else gm.YoutubeDnn
    dm = cls(U, T, D, D, Cc, att=att) if kind == 0 else cls(U, T, D, D, Cc)
    dm.set_weights("mlp0", om.W0); dm.set_weights("mlp1", om.W1); dm.set_weights("mlp2", om.W2)
    if kind == 0:
        dm.set_weights("att0", om.att0)
    return om, dm, SampleInfo.from_dims(U, T, D, Cc)


DIMS = [(5, 3, 7, 5), (52, 10, 16, 53), (52, 50, 16, 53), (52, 50, 64, 53)]   # last: BASELINE cfg4 (D=64, I=233)


def test_device_is_mi355x():
    from goctr_amd import capi
    capi.init()
    name, cus, hbm = capi.device_info()
    assert "gfx950" in name and cus >= 200 and hbm > 200e9


def test_weights_roundtrip_and_marshal_json(oracle):
    import json
    from goctr_amd import model as gm
    rng = np.random.default_rng(0)
    om, dm, si = pair(oracle, 0, 5, 3, 7, 5, rng)
    for n, ref in (("mlp0", om.W0), ("mlp1", om.W1), ("mlp2", om.W2), ("att0", om.att0.reshape(1, -1))):
        assert np.array_equal(dm.get_weights(n), ref)
    d = json.loads(dm.Marshal())                      # din.go:41-52 field names
    assert set(d) == {"uProfileDim", "uBehaviorSize", "uBehaviorDim", "iFeatureDim", "cFeatureDim", "mlp0", "mlp1",
                      "mlp2", "att0"}
    dm2 = gm.NewDinNetFromJson(dm.Marshal())
    assert np.array_equal(dm2.get_weights("mlp0"), om.W0) and np.array_equal(dm2.get_weights("att0"), om.att0.reshape(1, -1))
    with pytest.raises(ValueError):
        gm.NewDinNet(5, 3, 7, 8, 5)                   # din.go:176-178


@pytest.mark.parametrize("D,T", [(16, 10), (7, 3), (64, 50)])
def test_gather_bit_exact(oracle, D, T):
    from goctr_amd import model as gm
    rng = np.random.default_rng(1)
    V, U, Cc, rows = 1000, 5, 6, 777
    emb = rng.standard_normal((V, D)).astype(np.float32)
    ub = rng.integers(-2, V + 3, size=(rows, T)).astype(np.int32)      # includes missing / out-of-range ids
    it = rng.integers(-2, V + 3, size=rows).astype(np.int32)
    uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
    tab = gm.EmbeddingTable(emb)
    X = tab.gather_rows(ub, it, uf, cf)
    assert np.array_equal(X, oracle.assemble_rows(emb, ub, it, uf, cf))  # bit-exact


@pytest.mark.parametrize("kind,att", [(0, 0), (0, 1), (1, 0)])
@pytest.mark.parametrize("dims", DIMS)
def test_predict_logits(oracle, kind, att, dims):
    from goctr_amd import model as gm
    U, T, D, Cc = dims
    rng = np.random.default_rng(2)
    om, dm, si = pair(oracle, kind, U, T, D, Cc, rng, att=att)
    X, _ = make_data(rng, 118, U, T, D, Cc)
    gm.InitForwardOnlyVm(U, T, D, D, Cc, 20, dm)
    y = gm.Predict(dm, 118, 20, si, X)                # 118 rows @ batch 20: model_test.go:33-34
    ref = om.predict(X, 20)
    assert y.shape == (118,)
    assert np.max(np.abs(y - ref)) <= LOGIT_TOL


@pytest.mark.parametrize("kind,att", [(0, 0), (0, 1), (1, 0)])
@pytest.mark.parametrize("dims,B,valid", [((5, 3, 7, 5), 8, 8), ((52, 10, 16, 53), 200, 137), ((52, 50, 16, 53), 512, 512),
                                          ((52, 50, 64, 53), 256, 250)])
def test_loss_and_grads(oracle, kind, att, dims, B, valid):
    from goctr_amd import model as gm
    U, T, D, Cc = dims
    rng = np.random.default_rng(3)
    om, dm, si = pair(oracle, kind, U, T, D, Cc, rng, att=att, scale=0.15)
    X, Y = make_data(rng, valid, U, T, D, Cc)
    cost, g, y = gm.loss_grad(dm, si, X, Y, B=B)
    rcost, rg, ry = om.loss_grad(X, Y, B=B)           # padded rows count (quirk Q3)
    assert np.max(np.abs(y - ry)) <= LOGIT_TOL
    assert abs(cost - rcost) <= LOSS_TOL
    c64, g64, y64 = om.loss_grad_f64(X, Y, B=B)       # float64 truth of the same graph
    assert abs(cost - c64) <= LOSS_TOL and np.max(np.abs(y - y64)) <= LOGIT_TOL
    assert grad_bound(g["mlp0"], rg["W0"], g64["W0"]) and grad_bound(g["mlp1"], rg["W1"], g64["W1"])
    assert grad_bound(g["mlp2"], rg["W2"], g64["W2"])
    if kind == 0:
        assert grad_bound(g["att0"], rg["att0"], g64["att0"])


def test_loss_and_grads_reference_init(oracle):
    """with the reference's unscaled N(0,1) init most sigmoids saturate; logits/loss must still agree"""
    from goctr_amd import model as gm
    U, T, D, Cc = 52, 10, 16, 53
    rng = np.random.default_rng(4)
    om, dm, si = pair(oracle, 0, U, T, D, Cc, rng)
    X, Y = make_data(rng, 200, U, T, D, Cc)
    cost, g, y = gm.loss_grad(dm, si, X, Y)
    rcost, rg, ry = om.loss_grad(X, Y)
    assert np.max(np.abs(y - ry)) <= LOGIT_TOL
    if np.isfinite(rcost):
        assert abs(cost - rcost) <= 1e-5 * max(1.0, abs(rcost))
    else:
        assert not np.isfinite(cost)


def test_dropout_injected_mask_and_hash(oracle):
    from goctr_amd import capi, model as gm
    U, T, D, Cc, B = 52, 10, 16, 53, 64
    rng = np.random.default_rng(5)
    om, dm, si = pair(oracle, 0, U, T, D, Cc, rng, scale=0.15)
    X, Y = make_data(rng, B, U, T, D, Cc)
    m0 = (rng.random((B, 200)) < 0.75).astype(np.float32); m1 = (rng.random((B, 80)) < 0.5).astype(np.float32)
    cfg = capi.default_train_cfg(batch=B, dropout_mode=1, p0=0.25, p1=0.5)
    cost, g, y = gm.loss_grad(dm, si, X, Y, cfg=cfg, m0=m0, m1=m1)
    rcost, rg, ry = om.loss_grad(X, Y, drop=dict(mode=1, p0=0.25, p1=0.5, m0=m0, m1=m1))
    assert np.max(np.abs(y - ry)) <= LOGIT_TOL and abs(cost - rcost) <= LOSS_TOL
    assert grad_close(g["mlp0"], rg["W0"]) and grad_close(g["mlp1"], rg["W1"])
    # counter-hash mask: identical bits on host and device
    cfg = capi.default_train_cfg(batch=B, dropout_mode=2, p0=0.25, p1=0.5, seed=1234)
    cost, g, y = gm.loss_grad(dm, si, X, Y, cfg=cfg, step=7)
    rcost, rg, ry = om.loss_grad(X, Y, drop=dict(mode=2, p0=0.25, p1=0.5, seed=1234, step=7))
    assert np.max(np.abs(y - ry)) <= LOGIT_TOL and abs(cost - rcost) <= LOSS_TOL
    assert grad_close(g["mlp0"], rg["W0"]) and grad_close(g["mlp1"], rg["W1"]) and grad_close(g["att0"].ravel(), rg["att0"])


@pytest.mark.parametrize("kind", [0, 1])
def test_train_epochs_match_oracle(oracle, kind):
    """model.Train semantics: padded last batch, Adam per batch, cost of the last batch per epoch."""
    from goctr_amd import model as gm
    U, T, D, Cc = 5, 3, 7, 5
    rng = np.random.default_rng(6)
    om, dm, si = pair(oracle, kind, U, T, D, Cc, rng, scale=0.3)
    X, Y = make_data(rng, 250, U, T, D, Cc)
    ref = om.train(X, Y, batch=64, epochs=3)                      # 4 batches/epoch, last one padded
    costs = gm.Train(U, T, D, D, Cc, 250, 64, 3, 0, si, X, Y.reshape(-1, 1), dm)
    assert len(costs) == len(ref) == 3
    assert np.max(np.abs(costs - ref)) <= COST_TOL_SMALL_SHAPES
    for n, r in (("mlp0", om.W0), ("mlp1", om.W1), ("mlp2", om.W2)):
        assert np.max(np.abs(dm.get_weights(n) - r)) <= 2e-4, n     # 12 Adam steps of lr 0.01
    # after training the predictions still agree
    assert np.max(np.abs(gm.Predict(dm, 250, 100, si, X) - om.predict(X, 100))) <= 1e-4


def test_early_stop_matches_oracle(oracle):
    from goctr_amd import model as gm
    U, T, D, Cc = 5, 3, 7, 5
    rng = np.random.default_rng(7)
    om, dm, si = pair(oracle, 1, U, T, D, Cc, rng)
    X, Y = make_data(rng, 64, U, T, D, Cc)
    ref = om.train(X, Y, batch=32, epochs=40, early_stop=2)
    costs = gm.Train(U, T, D, D, Cc, 64, 32, 40, 2, si, X, Y, dm)
    assert len(costs) == len(ref)


def test_train_with_hash_dropout_matches_oracle(oracle):
    from goctr_amd import capi, model as gm
    U, T, D, Cc = 5, 3, 7, 5
    rng = np.random.default_rng(8)
    om, dm, si = pair(oracle, 0, U, T, D, Cc, rng, scale=0.3)
    X, Y = make_data(rng, 128, U, T, D, Cc)
    ref = om.train(X, Y, batch=32, epochs=2, drop_mode=2, p0=0.1, p1=0.2, seed=99)
    ds = gm.Dataset.dense(X, Y, si)
    cfg = capi.default_train_cfg(batch=32, epochs=2, early_stop=0, dropout_mode=2, p0=0.1, p1=0.2, seed=99)
    costs = gm.train_dataset(dm, ds, cfg)
    assert np.max(np.abs(costs - ref)) <= COST_TOL_SMALL_SHAPES


@pytest.mark.parametrize("kind", [0, 1])
def test_id_mode_equals_dense_mode(oracle, kind):
    """performance mode (ids + table in HBM) must compute the same function as the TrainSample rows"""
    from goctr_amd import capi, model as gm
    U, T, D, Cc, V, rows, B = 52, 50, 16, 53, 5000, 1024, 256
    rng = np.random.default_rng(9)
    om, dm, si = pair(oracle, kind, U, T, D, Cc, rng, scale=0.15)
    emb = (rng.standard_normal((V, D)) * 0.5).astype(np.float32)
    ub = rng.integers(0, V, size=(rows, T)).astype(np.int32)
    ub[rng.random((rows, T)) < 0.2] = -1                                  # 20 % padding slots
    it = rng.integers(0, V, size=rows).astype(np.int32)
    uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
    Y = (rng.random(rows) < 0.5).astype(np.float32)
    X = oracle.assemble_rows(emb, ub, it, uf, cf)
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, Y)
    y = gm.predict_dataset(dm, ds, B, emb=tab)
    assert np.max(np.abs(y - om.predict(X, B))) <= LOGIT_TOL
    cfg = capi.default_train_cfg(batch=B, epochs=2, early_stop=0, dropout_mode=0)
    costs = gm.train_dataset(dm, ds, cfg, emb=tab)
    ref = om.train(X, Y, batch=B, epochs=2)
    assert np.max(np.abs(costs - ref)) <= COST_TOL_SMALL_SHAPES


@pytest.mark.parametrize("steps", [7, 39])     # 4 + 2 + 1 and 16 + 16 + 4 + 2 + 1 steps per graph launch (ctr.hip run_steps)
def test_graph_replay_equals_eager(oracle, steps):
    from goctr_amd import capi, model as gm
    U, T, D, Cc = 52, 10, 16, 53
    rng = np.random.default_rng(10)
    X, Y = make_data(rng, 1000, U, T, D, Cc)
    res = []
    for no_graph in ("0", "1"):
        os.environ["GOCTR_NO_GRAPH"] = no_graph
        om, dm, si = pair(oracle, 0, U, T, D, Cc, np.random.default_rng(11), scale=0.15)
        ds = gm.Dataset.dense(X, Y, si)
        cfg = capi.default_train_cfg(batch=200, epochs=1, dropout_mode=0)
        costs = gm.train_steps(dm, ds, cfg, steps, want_costs=True)
        res.append((costs, dm.get_weights("mlp0")))
    os.environ.pop("GOCTR_NO_GRAPH")
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])   # deterministic


def test_full_size_properties_cfg3():
    """BASELINE config 3 shapes (DIN cosine, T=50, B=8192): properties that need no oracle run --
    determinism, zero-behaviour rows pool to zero, loss decreases on a learnable rule."""
    from goctr_amd import capi, model as gm
    from goctr_amd.recommend import SampleInfo
    U, T, D, Cc, V, rows, B = 52, 50, 16, 53, 26744, 1 << 15, 8192
    rng = np.random.default_rng(12)
    emb = (rng.standard_normal((V, D)) * 0.3).astype(np.float32)
    ub = rng.integers(0, V, size=(rows, T)).astype(np.int32)
    ub[rng.random((rows, T)) < 0.2] = -1
    it = rng.integers(0, V, size=rows).astype(np.int32)
    uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
    Y = (uf[:, 0] + cf[:, 0] > 1.0).astype(np.float32)                     # learnable rule
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, Y)

    def run():
        m = gm.DinNet(U, T, D, D, Cc)
        r = np.random.default_rng(13)
        m.set_weights("mlp0", (r.standard_normal((U + 2 * D + Cc, 200)) * 0.1).astype(np.float32))
        m.set_weights("mlp1", (r.standard_normal((200, 80)) * 0.1).astype(np.float32))
        m.set_weights("mlp2", (r.standard_normal((80, 1)) * 0.1).astype(np.float32))
        cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0)
        costs = gm.train_steps(m, ds, cfg, 40, emb=tab, want_costs=True)
        return costs, m.get_weights("mlp0"), gm.predict_dataset(m, ds, 4096, emb=tab)

    c1, w1, y1 = run()
    c2, w2, y2 = run()
    assert np.array_equal(c1, c2) and np.array_equal(w1, w2) and np.array_equal(y1, y2)   # bitwise reproducible
    assert np.all(np.isfinite(c1)) and c1[-1] < c1[0]                                       # learns
    assert np.all((y1 > 0) & (y1 < 1))
    from sklearn.metrics import roc_auc_score
    assert roc_auc_score(Y, y1) > 0.6


def test_modular_path_equals_fused_chain(oracle):
    """GOCTR_NO_CHAIN=1 (generic per-layer GEMM launches, used for hidden widths other than 200/80) must agree
    with the fused chain kernel"""
    from goctr_amd import capi, model as gm
    U, T, D, Cc = 52, 10, 16, 53
    rng = np.random.default_rng(14)
    X, Y = make_data(rng, 600, U, T, D, Cc)
    res = []
    for no_chain in ("0", "1"):
        os.environ["GOCTR_NO_CHAIN"] = no_chain
        om, dm, si = pair(oracle, 0, U, T, D, Cc, np.random.default_rng(15), scale=0.15)
        ds = gm.Dataset.dense(X, Y, si)
        cfg = capi.default_train_cfg(batch=200, epochs=1, dropout_mode=2, p0=0.1, p1=0.1, seed=5)
        costs = gm.train_steps(dm, ds, cfg, 6, want_costs=True)
        res.append((costs, dm.get_weights("mlp0"), dm.get_weights("att0")))
    os.environ.pop("GOCTR_NO_CHAIN")
    assert np.max(np.abs(res[0][0] - res[1][0])) <= 2e-5
    assert np.max(np.abs(res[0][1] - res[1][1])) <= 1e-4 and np.max(np.abs(res[0][2] - res[1][2])) <= 1e-4


@pytest.mark.parametrize("kind,att", [(0, 0), (0, 1), (1, 0)])
@pytest.mark.parametrize("U,T,D,Cc", [
    (3, 5, 4, 2),        # one lane per row (LPR 1), tiny everything
    (9, 17, 8, 11),      # LPR 2
    (52, 50, 16, 53),    # the cfg3 shape (LPR 4)
    (20, 70, 32, 10),    # LPR 8, T > 64: two id blocks per sample
    (52, 50, 64, 53),    # the cfg4 shape (LPR 16)
    (7, 12, 12, 5),      # D % 4 == 0 but not a power-of-two lane count: run-time mode kernel
    (6, 9, 10, 4),       # D % 4 != 0: scalar-lane kernel
    (140, 20, 16, 150),  # side-feature blocks wider than 128 columns
])
def test_id_mode_shape_sweep(oracle, kind, att, U, T, D, Cc):
    """every attention-kernel instantiation (vector width, lanes per row, compile-time mode) against the oracle"""
    from goctr_amd import capi, model as gm
    V, rows, B = 300, 192, 64
    rng = np.random.default_rng(U + T + D + Cc + 10 * kind + att)
    om, dm, si = pair(oracle, kind, U, T, D, Cc, rng, att=att, scale=0.15)
    emb = (rng.standard_normal((V, D)) * 0.5).astype(np.float32)
    ub = rng.integers(0, V, size=(rows, T)).astype(np.int32)
    ub[rng.random((rows, T)) < 0.25] = -1
    ub[0, :] = -1                                                         # a sample with no behaviour at all
    it = rng.integers(0, V, size=rows).astype(np.int32)
    it[1] = -1                                                            # a missing candidate embedding
    uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
    Y = (rng.random(rows) < 0.5).astype(np.float32)
    X = oracle.assemble_rows(emb, ub, it, uf, cf)
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, Y)
    y = gm.predict_dataset(dm, ds, B, emb=tab)
    assert np.max(np.abs(y - om.predict(X, B))) <= LOGIT_TOL
    cfg = capi.default_train_cfg(batch=B, epochs=2, early_stop=0, dropout_mode=0)
    costs = gm.train_dataset(dm, ds, cfg, emb=tab)
    ref = om.train(X, Y, batch=B, epochs=2)
    assert np.max(np.abs(costs - ref)) <= COST_TOL_SMALL_SHAPES
    for name, w in (("mlp0", om.W0), ("mlp1", om.W1), ("mlp2", om.W2)):
        assert np.max(np.abs(dm.get_weights(name) - w)) <= 2e-4 * max(1.0, np.max(np.abs(w)))


@pytest.mark.parametrize("att", [0, 1])
@pytest.mark.parametrize("T,B,rows", [(50, 100, 350), (64, 64, 200), (7, 33, 99), (1, 32, 64)])
def test_attention_backward_in_the_chain_kernel(oracle, att, T, B, rows):
    """DIN, D = 16, frozen embeddings, id mode: the att0 gradient's per-sample terms come out of ctr_chain_x3_kernel's tail
    (ChainX3Args::ab_*).  Batches that are not a multiple of the 32-row tile, a ragged last batch (rows past the dataset's
    end), T = 64 (every slot lane busy) and T = 1, missing ids: att0 after 2 epochs against the oracle, and bit for bit
    against the separate attn_bwd_kernel (GOCTR_CHAIN_ATTN_BWD=0).  Round 6: by default the chain launch also SUMS the terms over
    each tile (and dW2 likewise: ChainX3Args::tile_att0 / tile_dw2) -- another float32 summation order than the ones-column
    product of the weight-gradient launch, so that default is held to the oracle and to 2e-6 of the stored-terms path, and the
    bit-for-bit comparison is between the two stored-terms paths (GOCTR_CHAIN_TILE_SUMS=0)."""
    from goctr_amd import capi, model as gm
    U, D, Cc, V = 52, 16, 53, 300
    rng = np.random.default_rng(1000 + 10 * T + att)
    emb = (rng.standard_normal((V, D)) * 0.5).astype(np.float32)
    ub = rng.integers(0, V, size=(rows, T)).astype(np.int32)
    ub[rng.random((rows, T)) < 0.25] = -1
    ub[0, :] = -1
    it = rng.integers(0, V, size=rows).astype(np.int32)
    uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
    Y = (rng.random(rows) < 0.5).astype(np.float32)
    X = oracle.assemble_rows(emb, ub, it, uf, cf)
    res = []
    for sums, knob in (("1", None), ("0", None), ("0", "0")):
        os.environ["GOCTR_CHAIN_TILE_SUMS"] = sums
        if knob is not None:
            os.environ["GOCTR_CHAIN_ATTN_BWD"] = knob
        try:
            om, dm, si = pair(oracle, 0, U, T, D, Cc, np.random.default_rng(77), att=att, scale=0.15)
            tab = gm.EmbeddingTable(emb)
            ds = gm.Dataset.ids(ub, it, uf, cf, Y)
            cfg = capi.default_train_cfg(batch=B, epochs=2, early_stop=0, dropout_mode=2, p0=0.01, p1=0.01, seed=9)
            costs = gm.train_dataset(dm, ds, cfg, emb=tab)
            res.append((costs, dm.get_weights("att0"), dm.get_weights("mlp0"), dm.get_weights("mlp1")))
        finally:
            os.environ.pop("GOCTR_CHAIN_ATTN_BWD", None)
            os.environ.pop("GOCTR_CHAIN_TILE_SUMS", None)
    for a, b in zip(res[1], res[2]):
        assert np.array_equal(a, b)
    for a, b in zip(res[0], res[1]):               # per-tile sums against stored terms: float32 summation order only
        assert np.max(np.abs(a - b)) <= 2e-6 * max(1.0, float(np.max(np.abs(b))))
    ref = om.train(X, Y, batch=B, epochs=2, drop_mode=2, p0=0.01, p1=0.01, seed=9)
    assert np.max(np.abs(res[0][0] - ref)) <= COST_TOL_SMALL_SHAPES
    assert np.max(np.abs(res[0][1].ravel() - om.att0.ravel())) <= 2e-4
    assert np.max(np.abs(res[0][2] - om.W0)) <= 2e-4 * max(1.0, np.max(np.abs(om.W0)))


def test_graph_is_rebuilt_for_a_new_dataset_at_a_reused_address(oracle):
    """the cached step graph bakes in the dataset's device pointers and row count: it must be keyed on the dataset's
    generation, not on the host address of its handle (malloc readily returns a freed handle's address).  Train on
    dataset A, destroy it, create B (different size) -- repeatedly, so that an address is reused -- and compare every
    run with the eager path"""
    from goctr_amd import capi, model as gm
    U, T, D, Cc, B = 52, 10, 16, 53, 128
    rng = np.random.default_rng(21)
    sets = [make_data(rng, rows, U, T, D, Cc) for rows in (640, 384, 896, 384, 640, 256)]
    res = []
    for no_graph in ("0", "1"):
        os.environ["GOCTR_NO_GRAPH"] = no_graph
        om, dm, si = pair(oracle, 0, U, T, D, Cc, np.random.default_rng(22), scale=0.15)
        cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0)
        costs, handles = [], set()
        for X, Y in sets:
            ds = gm.Dataset.dense(X, Y, si)
            handles.add(ds._h.value)
            costs.append(gm.train_steps(dm, ds, cfg, 4, want_costs=True))
            ds.close()                                      # goctr_dataset_destroy: the next create may reuse the address
        res.append((np.concatenate(costs), dm.get_weights("mlp0"), len(handles)))
    os.environ.pop("GOCTR_NO_GRAPH")
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    # and against the oracle: the same sequence of 24 steps
    om, dm, si = pair(oracle, 0, U, T, D, Cc, np.random.default_rng(22), scale=0.15)
    ref = []
    st = None
    for X, Y in sets:
        for b in range(4):
            nb = -(-X.shape[0] // B)
            lo = (b % nb) * B
            c, g, _ = om.loss_grad(X[lo:lo + B], Y[lo:lo + B], B=B)
            st = om.adam_step(g, state=st, batch=B)
            ref.append(c)
    assert np.max(np.abs(res[0][0] - np.array(ref, np.float32))) <= COST_TOL_SMALL_SHAPES


def test_concurrent_handles_from_several_threads(oracle):
    """include/goctr.h: any host thread may call any handle (goroutines of a cgo host; PredictAbstract.Predict is called
    concurrently from gin handlers, recommend/api.go:106-131).  Two threads predict on their own models while a third
    trains a third model (captures and replays step graphs on the engine's stream): every result equals the
    single-threaded one"""
    import threading
    from goctr_amd import capi, model as gm
    U, T, D, Cc = 52, 10, 16, 53
    rng = np.random.default_rng(23)
    X, Y = make_data(rng, 2000, U, T, D, Cc)
    pairs = [pair(oracle, k % 2, U, T, D, Cc, np.random.default_rng(30 + k), scale=0.15) for k in range(3)]
    si = pairs[0][2]
    want_pred = [gm.Predict(pairs[k][1], 2000, 256, si, X) for k in range(2)]
    ds = gm.Dataset.dense(X, Y, si)
    cfg = capi.default_train_cfg(batch=200, epochs=1, dropout_mode=0)
    # single-threaded reference for the trainer: a twin of model 2
    twin = pair(oracle, 0, U, T, D, Cc, np.random.default_rng(32), scale=0.15)[1]
    want_costs = np.concatenate([gm.train_steps(twin, ds, cfg, 5, first_batch=5 * r, want_costs=True) for r in range(6)])
    out, errs = {}, []

    def predictor(k):
        try:
            out[k] = [gm.Predict(pairs[k][1], 2000, 256, si, X) for _ in range(12)]
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    def trainer():
        try:
            out["t"] = np.concatenate([gm.train_steps(pairs[2][1], ds, cfg, 5, first_batch=5 * r, want_costs=True) for r in range(6)])
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=predictor, args=(0,)), threading.Thread(target=predictor, args=(1,)),
          threading.Thread(target=trainer)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(2):
        assert all(np.array_equal(y, want_pred[k]) for y in out[k])
    assert np.array_equal(out["t"], want_costs)
    assert np.array_equal(pairs[2][1].get_weights("mlp0"), twin.get_weights("mlp0"))


@pytest.mark.parametrize("kind", [0, 1])
def test_grouped_predict_launches_give_identical_scores(oracle, kind):
    """goctr_predict_* scores GOCTR_PRED_GROUP consecutive PredBatchSize batches per launch: every row is scored whatever the
    grouping (rows not a multiple of the batch, a batch count not a multiple of the group), and the scores agree to float32
    rounding -- the grouped launches take the 32-row-tile forward kernel, single batches the 16-row-tile one"""
    from goctr_amd import capi, model as gm
    U, T, D, Cc, V, rows, PB = 52, 50, 16 if kind == 0 else 64, 53, 3000, 4096 * 5 + 777, 4096
    rng = np.random.default_rng(31)
    emb = (rng.standard_normal((V, D)) * 0.3).astype(np.float32)
    ub = rng.integers(-1, V, size=(rows, T)).astype(np.int32)
    it = rng.integers(0, V, size=rows).astype(np.int32)
    uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, None)
    m = (gm.DinNet if kind == 0 else gm.YoutubeDnn)(U, T, D, D, Cc)
    r = np.random.default_rng(32)
    m.set_weights("mlp0", (r.standard_normal((U + 2 * D + Cc, 200)) * 0.1).astype(np.float32))
    m.set_weights("mlp1", (r.standard_normal((200, 80)) * 0.1).astype(np.float32))
    m.set_weights("mlp2", (r.standard_normal((80, 1)) * 0.1).astype(np.float32))
    res = []
    for g in ("1", "4", "3"):
        os.environ["GOCTR_PRED_GROUP"] = g
        res.append(gm.predict_dataset(m, ds, PB, emb=tab))
    os.environ.pop("GOCTR_PRED_GROUP")
    assert res[0].shape == (rows,) and np.all((res[0] > 0) & (res[0] < 1))
    assert np.max(np.abs(res[0] - res[1])) <= 2e-6 and np.max(np.abs(res[0] - res[2])) <= 2e-6
    os.environ["GOCTR_PRED_GROUP"] = "4"
    again = gm.predict_dataset(m, ds, PB, emb=tab)
    os.environ.pop("GOCTR_PRED_GROUP")
    assert np.array_equal(again, res[1])                                         # the same grouping: the same bits


def test_predict_between_training_calls_leaves_training_untouched(oracle):
    """a predict whose (grouped) launches need a larger workspace reallocates it and drops the captured step graphs: the
    next training call must rebuild them and land on the same bits as a run without the predict in between"""
    from goctr_amd import capi, model as gm
    U, T, D, Cc, V, rows, B = 52, 50, 16, 53, 2000, 2000, 200
    rng = np.random.default_rng(41)
    emb = (rng.standard_normal((V, D)) * 0.3).astype(np.float32)
    ub = rng.integers(-1, V, size=(rows, T)).astype(np.int32)
    it = rng.integers(0, V, size=rows).astype(np.int32)
    uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
    Y = (uf[:, 0] > 0.5).astype(np.float32)
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, Y)
    res = []
    for interrupt in (False, True):
        m = gm.DinNet(U, T, D, D, Cc)
        r = np.random.default_rng(42)
        m.set_weights("mlp0", (r.standard_normal((U + 2 * D + Cc, 200)) * 0.1).astype(np.float32))
        m.set_weights("mlp1", (r.standard_normal((200, 80)) * 0.1).astype(np.float32))
        m.set_weights("mlp2", (r.standard_normal((80, 1)) * 0.1).astype(np.float32))
        cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2, p0=0.005, p1=0.005, seed=7)
        gm.train_steps(m, ds, cfg, 5, emb=tab)
        if interrupt:
            y = gm.predict_dataset(m, ds, 4096, emb=tab)
            assert y.shape == (rows,) and np.all(np.isfinite(y))
        gm.train_steps(m, ds, cfg, 6, first_batch=5, emb=tab)
        res.append((m.get_weights("mlp0"), m.get_weights("att0")))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


