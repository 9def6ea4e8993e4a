"""GPU parity tests for the sklearn-port MLP path (float64 on v_mfma_f64_16x16x4_f64) against the CPU
oracle (oracle/orc_sklmlp.c).  Tolerance: 1e-9 relative -- both sides are float64 and differ only in
summation order (and pow() vs the reference's running beta products in the Adam quirk Q7)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def close(a, b, rtol=RTOL, atol=1e-13):
    return np.allclose(a, b, rtol=rtol, atol=atol)


def make(rng, n, F):
    X = rng.random((n, F)).astype(np.float32)
    Y = ((X[:, 0] + X[:, 1]) > 1).astype(np.float32).reshape(-1, 1)
    return X, Y


@pytest.mark.parametrize("act", ["relu", "logistic", "tanh", "identity"])
@pytest.mark.parametrize("units", [[6, 4, 1], [281, 100, 1], [40, 33, 17, 1]])
def test_loss_grad(oracle, act, units):
    from goctr_amd import mlp as gmlp
    rng = np.random.default_rng(0)
    clf = gmlp.MLPClassifier(units[1:-1], act, "adam", 1e-2)
    theta = clf.init_params(units, rng) * 0.5
    clf.create(units, 64, theta)
    X, Y = make(rng, 64, units[0])
    loss, g = clf.loss_grad(X, Y)
    cfg = oracle.mlp_cfg(units, act, alpha=1e-2)
    rloss, rg = oracle.mlp_loss_grad(cfg, theta.copy(), X.astype(np.float64), Y.astype(np.float64))
    assert loss == pytest.approx(rloss, rel=RTOL)
    assert close(g, rg)


def test_predict(oracle):
    from goctr_amd import mlp as gmlp
    rng = np.random.default_rng(1)
    units = [281, 100, 1]
    clf = gmlp.MLPClassifier([100], "relu", "adam", 1e-5)
    theta = clf.init_params(units, rng)
    clf.create(units, 200, theta)
    X, _ = make(rng, 1237, 281)
    y = clf.Predict(X)
    ref = oracle.mlp_predict(oracle.mlp_cfg(units, "relu", 1e-5), theta, X.astype(np.float64))
    assert y.dtype == np.float32 and y.shape == (1237, 1)
    assert np.array_equal(y, ref.astype(np.float32)) or np.max(np.abs(y - ref)) < 1e-7


@pytest.mark.parametrize("solver", ["adam", "sgd"])
def test_fit_matches_oracle(oracle, solver):
    """fitStochastic with a given batch order: loss curve + final parameters (incl. the Adam quirk Q7)"""
    from goctr_amd import mlp as gmlp
    rng = np.random.default_rng(2)
    units = [20, 12, 1]
    n, batch, iters = 600, 200, 6
    X, Y = make(rng, n, 20)
    clf = gmlp.MLPClassifier([12], "relu", solver, 1e-4)
    clf.BatchSize, clf.MaxIter, clf.Tol = batch, iters, -1.0          # never stop early
    theta0 = clf.init_params(units, rng)
    perm = np.stack([rng.permutation(n) for _ in range(iters)]).astype(np.int32)
    clf.Fit(X, Y, theta0=theta0.copy(), perm=perm)
    cfg = oracle.mlp_cfg(units, "relu", alpha=1e-4)
    theta = theta0.copy()
    opt = oracle.MlpOptimizer(solver, theta.size)
    ref = oracle.mlp_fit(cfg, theta, opt, X.astype(np.float64), Y.astype(np.float64), batch, iters, tol=-1.0, perm=perm)
    assert close(clf.LossCurve, ref, rtol=1e-8)
    assert close(clf.get_params(), theta, rtol=1e-7, atol=1e-10)


def test_fit_stops_on_tolerance(oracle):
    from goctr_amd import mlp as gmlp
    rng = np.random.default_rng(3)
    units = [8, 6, 1]
    X, Y = make(rng, 400, 8)
    clf = gmlp.MLPClassifier([6], "relu", "adam", 1e-5)
    clf.BatchSize, clf.MaxIter, clf.Tol, clf.Shuffle = 200, 60, 1e-2, False
    theta0 = clf.init_params(units, rng)
    clf.Fit(X, Y, theta0=theta0.copy())
    theta = theta0.copy()
    ref = oracle.mlp_fit(oracle.mlp_cfg(units, "relu", 1e-5), theta, oracle.MlpOptimizer("adam", theta.size),
                         X.astype(np.float64), Y.astype(np.float64), 200, 60, tol=1e-2)
    assert clf.NIter == len(ref) < 60


def test_batchnorm_and_weight_decay(oracle):
    from goctr_amd import mlp as gmlp
    rng = np.random.default_rng(4)
    units = [10, 7, 1]
    clf = gmlp.MLPClassifier([7], "relu", "adam", 1e-3)
    clf.BatchNormalize, clf.WeightDecay = True, 1e-3
    theta = clf.init_params(units, rng)
    clf.create(units, 50, theta)
    X, Y = make(rng, 50, 10)
    loss, g = clf.loss_grad(X, Y)
    cfg = oracle.mlp_cfg(units, "relu", alpha=1e-3, batch_normalize=True, weight_decay=1e-3)
    th = theta.copy()
    rloss, rg = oracle.mlp_loss_grad(cfg, th, X.astype(np.float64), Y.astype(np.float64))
    assert loss == pytest.approx(rloss, rel=RTOL) and close(g, rg)
    assert close(clf.get_params(), th)                      # theta *= (1 - wd) happened on both sides


def test_wrappers_and_hyperparameter_errors():
    from goctr_amd import mlp as gmlp
    from goctr_amd.recommend import SampleInfo, TrainSample
    rng = np.random.default_rng(5)
    X, Y = make(rng, 800, 12)
    clf = gmlp.NewMLPClassifier([16], "relu", "adam", 1e-5)       # main.go:39-46 style
    clf.MaxIter, clf.RandomState, clf.LearningRateInit = 200, np.random.default_rng(7), 0.01
    pred = gmlp.SimpleMlpFitWrap(clf).Fit(TrainSample(X.ravel(), Y.ravel(), 800, 12, SampleInfo()))
    p = pred.Predict(X)
    from sklearn.metrics import roc_auc_score
    assert roc_auc_score(Y.ravel(), p.ravel()) > 0.85
    assert clf.LossCurve[-1] < clf.LossCurve[0]
    bad = gmlp.MLPClassifier([4], "swish", "adam", 1e-5)
    with pytest.raises(ValueError):
        bad.Fit(X, Y)
    with pytest.raises(ValueError):
        gmlp.MLPClassifier([4], "relu", "lbfgs", 1e-5).Fit(X, Y)


def test_full_size_cfg2_properties():
    """BASELINE config 2 shape ([281,100,1], batch 4096): determinism and learning on a separable rule"""
    from goctr_amd import mlp as gmlp
    rng = np.random.default_rng(6)
    X, Y = make(rng, 1 << 15, 281)
    res = []
    for _ in range(2):
        clf = gmlp.MLPClassifier([100], "relu", "adam", 1e-5)
        units = [281, 100, 1]
        clf.create(units, 4096, clf.init_params(units, np.random.default_rng(8)))
        clf.upload(X, Y)
        clf.train_steps(64)
        res.append(clf.get_params())
    assert np.array_equal(res[0], res[1])
    l0, _ = clf.loss_grad(X[:4096], Y[:4096])
    clf.train_steps(200)
    l1, _ = clf.loss_grad(X[:4096], Y[:4096])
    assert np.isfinite(l1) and l1 < l0


def test_multi_step_graphs_equal_eager_steps():
    """goctr_mlp_train_steps replays graphs of 8 / 2 / 1 steps; the same 27 steps launched kernel by kernel must leave the
    same bits (both the straight-line F = 281 chain kernel and the run-time loop one)"""
    import os
    from goctr_amd import mlp as gmlp
    for F, H, B in ((281, 100, 512), (37, 12, 96)):
        rng = np.random.default_rng(16)
        X, Y = make(rng, 8 * B, F)
        res = []
        for no_graph in ("0", "1"):
            os.environ["GOCTR_NO_GRAPH"] = no_graph
            clf = gmlp.MLPClassifier([H], "relu", "adam", 1e-5)
            units = [F, H, 1]
            clf.create(units, B, clf.init_params(units, np.random.default_rng(17)))
            clf.upload(X, Y)
            clf.train_steps(27)
            res.append(clf.get_params())
        os.environ.pop("GOCTR_NO_GRAPH")
        assert np.array_equal(res[0], res[1])


def test_slabs_through_the_l2_leave_the_same_bits():
    """GOCTR_MLP_TN_WT=0 (plain slab stores in mlp_tn64_kernel) against the default (stores through the L2, sc1): where a store goes
    does not change what is stored"""
    import os
    from goctr_amd import capi, mlp as gmlp
    rng = np.random.default_rng(15)
    n, F, B = 4096, 281, 1024
    X = rng.random((n, F), dtype=np.float32)
    Y = (rng.random(n) < 0.5).astype(np.float32)
    units = [F, 100, 1]
    res = []
    for knob in (None, "0"):
        if knob is not None:
            os.environ["GOCTR_MLP_TN_WT"] = knob
        try:
            clf = gmlp.MLPClassifier([100], "relu", "adam", 1e-5)
            clf.create(units, B, clf.init_params(units, np.random.default_rng(3)))
            clf.upload(X, Y)
            clf.train_steps(11)
            capi.sync()
            res.append(clf.get_params())
        finally:
            os.environ.pop("GOCTR_MLP_TN_WT", None)
    assert np.array_equal(res[0], res[1])


@pytest.mark.parametrize("knob", ["GOCTR_MLP_X64", "GOCTR_MLP_PREFETCH"])
@pytest.mark.parametrize("B", [1024, 200])
def test_float64_row_image_and_prefetch_blocks_leave_the_same_bits(knob, B):
    """GOCTR_MLP_X64=0 (mlp_chain_kernel writes the float64 copy of its rows for the weight-gradient launch) against the default
    (the rows kept once as a float64 image, the chain launch writes row indices, mlp_tn64_kernel<3, true> reads through them), and
    GOCTR_MLP_PREFETCH=0 (no prefetch blocks behind the reduce launch): the same values enter the same products in the same order;
    with a permutation, over an epoch boundary (the prefetch of the last batch asks for batch 0 of the OLD permutation: harmless)"""
    import os
    from goctr_amd import capi, mlp as gmlp
    rng = np.random.default_rng(16)
    n, F = 4096 + 40, 281
    X = rng.random((n, F), dtype=np.float32)
    Y = (rng.random(n) < 0.5).astype(np.float32)
    units = [F, 100, 1]
    perm = np.stack([np.random.default_rng(5 + e).permutation(n).astype(np.int32) for e in range(2)])
    res = []
    for val in (None, "0"):
        if val is not None:
            os.environ[knob] = val
        try:
            clf = gmlp.MLPClassifier([100], "relu", "adam", 1e-5)
            clf.create(units, B, clf.init_params(units, np.random.default_rng(3)))
            clf.upload(X, Y)
            clf.train_steps(n // B + 3)
            capi.sync()
            p1 = clf.get_params()
            clf2 = gmlp.MLPClassifier([100], "relu", "adam", 1e-5)
            clf2.MaxIter, clf2.BatchSize = 2, B
            clf2.Fit(X, Y, theta0=clf2.init_params(units, np.random.default_rng(3)), perm=perm)   # two epochs, each with a short batch
            res.append((p1, clf2.get_params(), np.array(clf2.LossCurve)))
        finally:
            os.environ.pop(knob, None)
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def test_reference_nn_forward_kat():
    """the reference-held 3-3-3 forward vector (nn/network_test.go:25-83, tests/golden/ref_kats.json) through the device
    MLP: units [3,3,3], relu hidden, logistic output = the KAT's ReLU and Sigmoid layers; float32 out (mlp.go:33-38)"""
    import json, os
    from goctr_amd import mlp as gmlp
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kats.json")))["nn_forward"]
    parts = []
    for layer in k["weights_layer_neuron_input"][:2]:
        W = np.asarray(layer, np.float64).T
        parts += [np.full(W.shape[1], k["bias"]), W.ravel()]
    theta = np.concatenate(parts)
    clf = gmlp.MLPClassifier([3], "relu", "adam", 1e-4)
    clf.create([3, 3, 3], 8, theta)
    x = np.asarray([k["input"]] * 5, np.float32)
    y = clf.Predict(x)
    # inputs are narrowed to float32 at the boundary (0.1f, 0.2f, 0.7f), outputs too: a few float32 ulps
    assert y.shape == (5, 3) and np.max(np.abs(y - np.asarray(k["expected"][1]))) <= 2e-7
    # the float64 entry: loss/grad path sees the exact inputs -> compare the activations through the loss instead
    Y = np.asarray([[1.0, 0.0, 1.0]])
    loss, _ = clf.loss_grad(np.asarray([k["input"]], np.float64), Y)
    p = np.asarray(k["expected"][1])
    ref = -(Y * np.log(p) + (1 - Y) * np.log(1 - p)).sum() + 0.5 * 1e-4 * sum(
        (np.asarray(l, np.float64) ** 2).sum() for l in k["weights_layer_neuron_input"][:2])
    assert loss == pytest.approx(ref, rel=1e-9)


# ---- quirk Q11: the short last batch (basemlp64.go:790-812; oracle pinned by tests/test_oracle_mlp_q11.py) ----

@pytest.mark.parametrize("units,act,solver,bn,n,batch", [
    ([20, 12, 1], "relu", "adam", False, 600 + 148, 200),          # 3 whole batches + 148 rows
    ([20, 12, 1], "relu", "sgd", False, 200 + 1, 200),             # a ONE-row short batch
    ([281, 100, 1], "relu", "adam", False, 2 * 200 + 148, 200),    # the fused-chain shape (main.go:39-46) around the generic steps
    ([40, 33, 17, 1], "tanh", "adam", False, 3 * 64 + 63, 64),     # two hidden layers: only the first keeps stale rows
    ([40, 33, 17, 1], "logistic", "adam", False, 3 * 64 + 5, 64),
    ([10, 7, 1], "relu", "adam", True, 2 * 50 + 17, 50),           # max-abs normalisation runs over the stale rows too
    ([12, 9, 1], "identity", "adam", False, 96 + 31, 96),
])
def test_fit_trains_the_short_last_batch_like_the_reference(oracle, units, act, solver, bn, n, batch):
    from goctr_amd import mlp as gmlp
    rng = np.random.default_rng(21)
    iters = 4
    X, Y = make(rng, n, units[0])
    clf = gmlp.MLPClassifier(units[1:-1], act, solver, 1e-3)
    clf.BatchSize, clf.MaxIter, clf.Tol, clf.BatchNormalize = batch, iters, -1.0, bn
    theta0 = clf.init_params(units, rng)
    perm = np.stack([rng.permutation(n) for _ in range(iters)]).astype(np.int32)
    clf.Fit(X, Y, theta0=theta0.copy(), perm=perm)
    cfg = oracle.mlp_cfg(units, act, alpha=1e-3, batch_normalize=bn)
    theta = theta0.copy()
    opt = oracle.MlpOptimizer(solver, theta.size)
    ref = oracle.mlp_fit(cfg, theta, opt, X.astype(np.float64), Y.astype(np.float64), batch, iters, tol=-1.0, perm=perm)
    assert opt.o.t == iters * (n // batch + 1)                     # the short batch is an update of its own
    assert clf.NIter == iters and close(clf.LossCurve, ref, rtol=1e-8)
    assert close(clf.get_params(), theta, rtol=1e-7, atol=1e-10)
    # and it is NOT what dropping the tail computes (round 1-4 behaviour): the parameters differ visibly
    th2 = theta0.copy()
    nt = n // batch * batch
    keep = np.stack([p[np.isin(p, p[:nt])][:nt] for p in perm])   # any whole-batch-only schedule
    oracle.mlp_fit(cfg, th2, oracle.MlpOptimizer(solver, th2.size), X.astype(np.float64), Y.astype(np.float64), batch, iters,
                   tol=-1.0, perm=None)
    assert np.max(np.abs(th2 - theta)) > 1e-6 and keep.shape == (iters, nt)


def test_flagship_run_79948_rows_at_batch_200(oracle):
    """the reference's own entry point (main.go:39-50; BASELINE configs[0]): [281, 100, 1] relu / adam, alpha 1e-5, 79 948 rows
    (feature_test.go:24), BatchSize 200, MaxIter 20 -> 399 whole batches and one of 148 per epoch, a given shuffle.
    Loss curve 1e-8 relative and parameters 1e-7 against the oracle."""
    from goctr_amd import mlp as gmlp
    rng = np.random.default_rng(22)
    n, F, iters = 79948, 281, 20
    X = rng.random((n, F)).astype(np.float32)
    Y = ((X[:, :8].sum(1) + 0.3 * rng.standard_normal(n)) > 4).astype(np.float32).reshape(-1, 1)
    units = [F, 100, 1]
    clf = gmlp.MLPClassifier([100], "relu", "adam", 1e-5)
    clf.BatchSize, clf.MaxIter, clf.Tol = 200, iters, -1.0
    theta0 = clf.init_params(units, rng)
    order, perms = np.arange(n), []
    for _ in range(iters):                                         # cumulative in-place shuffles, like fitStochastic :786-788
        order = order[rng.permutation(n)]
        perms.append(order.copy())
    perm = np.stack(perms).astype(np.int32)
    clf.Fit(X, Y, theta0=theta0.copy(), perm=perm)
    cfg = oracle.mlp_cfg(units, "relu", alpha=1e-5)
    theta = theta0.copy()
    opt = oracle.MlpOptimizer("adam", theta.size)
    import os
    oracle.set_threads(min(16, len(os.sched_getaffinity(0))))      # (row-parallel loops: no summation order depends on it)
    try:
        ref = oracle.mlp_fit(cfg, theta, opt, X.astype(np.float64), Y.astype(np.float64), 200, iters, tol=-1.0, perm=perm)
    finally:
        oracle.set_threads(1)
    assert opt.o.t == iters * 400
    assert len(clf.LossCurve) == iters and close(clf.LossCurve, ref, rtol=1e-8)
    assert close(clf.get_params(), theta, rtol=1e-7, atol=1e-10)
    assert ref[-1] < ref[0]

