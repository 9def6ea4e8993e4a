"""Full-size parity: ONE step of every BASELINE configuration at the shape bench.py measures, against the CPU oracle.

The reduced-size tests (test_gpu_ctr.py, test_gpu_mlp.py) cover every code path; these cover the sizes at which the
launch geometry differs -- B = 8192 / 16384 rows means 256+ chain workgroups, 44 weight-gradient slabs and the
one-workgroup-per-CU sizing -- and the 2.56 GB table of cfg4.  Bars (BASELINE.json north_star): gather bit-exact,
logits / loss <= 1e-5 absolute.  Gradients are bounded against a FLOAT64 evaluation of the same graph
(pyoracle.CtrModel.loss_grad_f64): the device may be at most twice as far from the truth as the float32 oracle is --
the float32 oracle's own summation-order error is the yard-stick, not an ad-hoc tolerance.  That bound is what
entitles the 6-product bf16 split of the weight-gradient GEMM (csrc/mfma_gemm.h) to call itself float32.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-5
LOSS_TOL = 1e-5


def zipf_ids(rng, V, shape):
    return ((rng.zipf(1.05, size=shape) - 1) % V).astype(np.int32)


def synth(rng, rows, U, T, D, Cc, V, emb_scale=0.25):
    """bench.py's synthetic MovieLens-shaped keys: Zipf(1.05) ids, 20 % padded slots, U(0,1) side features"""
    ub = zipf_ids(rng, V, (rows, T))
    ub[rng.random((rows, T)) < 0.2] = -1
    it = zipf_ids(rng, V, rows)
    uf = rng.random((rows, U), dtype=np.float32)
    cf = rng.random((rows, Cc), dtype=np.float32)
    y = (rng.random(rows) < 0.5).astype(np.float32)
    emb = rng.random((V, D), dtype=np.float32)
    emb -= 0.5
    emb *= 2 * emb_scale
    return emb, ub, it, uf, cf, y


def models(oracle, kind, U, T, D, Cc, rng, scale):
    from goctr_amd import model as gm
    om = oracle.CtrModel(kind, U, T, D, Cc)
    om.W0[:] = (rng.standard_normal(om.W0.shape) * scale).astype(np.float32)
    om.W1[:] = (rng.standard_normal(om.W1.shape) * scale).astype(np.float32)
    om.W2[:] = (rng.standard_normal(om.W2.shape) * scale).astype(np.float32)
    if kind == 0:
        om.att0[:] = (1 + 0.3 * rng.standard_normal(T)).astype(np.float32)
    dm = (gm.DinNet if kind == 0 else gm.YoutubeDnn)(U, T, D, D, Cc)
    for n, w in (("mlp0", om.W0), ("mlp1", om.W1), ("mlp2", om.W2)):
        dm.set_weights(n, w)
    if kind == 0:
        dm.set_weights("att0", om.att0)
    return om, dm


def grad_bound(name, g_dev, g_orc, g64):
    """|g_dev - truth| <= 2 |g_orc32 - truth| (max norm, per tensor; + one float32 ulp of the largest entry)"""
    g_dev, g_orc, g64 = (np.asarray(a, np.float64).ravel() for a in (g_dev, g_orc, g64))
    e_dev, e_orc = np.max(np.abs(g_dev - g64)), np.max(np.abs(g_orc - g64))
    ulp = 6e-8 * np.max(np.abs(g64))
    assert e_dev <= 2 * e_orc + ulp, f"{name}: device {e_dev:.3e} from the float64 truth, float32 oracle {e_orc:.3e}"
    # and in the mean: the device must not be systematically worse either
    r_dev, r_orc = np.sqrt(np.mean((g_dev - g64) ** 2)), np.sqrt(np.mean((g_orc - g64) ** 2))
    assert r_dev <= 2 * r_orc + ulp, f"{name}: rms device {r_dev:.3e} vs oracle {r_orc:.3e}"
    return e_dev, e_orc


def one_step_checks(oracle, kind, U, T, D, Cc, V, B, seed, drop):
    from goctr_amd import capi, model as gm
    from goctr_amd.recommend import SampleInfo
    rng = np.random.default_rng(seed)
    emb, ub, it, uf, cf, Y = synth(rng, B, U, T, D, Cc, V)
    om, dm = models(oracle, kind, U, T, D, Cc, rng, 0.15)
    si = SampleInfo.from_dims(U, T, D, Cc)
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, Y)

    # gather: bit-exact, on a slice (the full [B, XCols] matrix is what the oracle assembles next anyway)
    X = oracle.assemble_rows(emb, ub, it, uf, cf)
    assert np.array_equal(tab.gather_rows(ub[:512], it[:512], uf[:512], cf[:512]), X[:512])

    # logits through the predict path at the bench's predict batch, id mode
    y = gm.predict_dataset(dm, ds, 4096, emb=tab)
    ry = om.predict(X, 4096)
    assert np.max(np.abs(y - ry)) <= LOGIT_TOL

    # loss + gradients of the full batch (dense rows: the parity entry), against oracle and float64 truth
    cost, g, yb = gm.loss_grad(dm, si, X, Y, B=B)
    rcost, rg, ryb = om.loss_grad(X, Y, B=B)
    c64, g64, y64 = om.loss_grad_f64(X, Y, B=B)
    assert np.max(np.abs(yb - ryb)) <= LOGIT_TOL and abs(cost - rcost) <= LOSS_TOL
    assert abs(cost - c64) <= LOSS_TOL and np.max(np.abs(yb - y64)) <= LOGIT_TOL
    pairs = [("mlp0", "W0"), ("mlp1", "W1"), ("mlp2", "W2")] + ([("att0", "att0")] if kind == 0 else [])
    for dn, on in pairs:
        grad_bound(on, g[dn], rg[on], g64[on])

    # Adam: the device's update of ITS gradient must be the oracle's AdamSolver.Step of that same gradient
    dsd = gm.Dataset.dense(X, Y, si)
    cfg0 = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0)
    c_dense = gm.train_steps(dm, dsd, cfg0, 1, want_costs=True)
    assert abs(c_dense[0] - rcost) <= LOSS_TOL
    ref = oracle.CtrModel(kind, U, T, D, Cc)
    for a in ("W0", "W1", "W2", "att0"):
        getattr(ref, a)[:] = getattr(om, a)
    ref.adam_step(dict(W0=g["mlp0"], W1=g["mlp1"], W2=g["mlp2"], att0=g["att0"].ravel()), batch=B)
    for dn, on in pairs:
        d = np.abs(dm.get_weights(dn).ravel() - getattr(ref, on).ravel())
        assert np.max(d) <= 2e-7, (dn, float(np.max(d)))

    # the step the bench times: id mode, graph-replayed, with the reference's dropout when asked; weights after the step
    # against the oracle's own step.  (An element whose total gradient passes within float32 noise of zero gets an
    # ill-conditioned first Adam update, +-lr: hence the quantile next to the max.)
    om2, dm2 = models(oracle, kind, U, T, D, Cc, np.random.default_rng(seed + 1), 0.15)
    cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2 if drop else 0, p0=0.005, p1=0.005, seed=77)
    c_id = gm.train_steps(dm2, ds, cfg, 1, emb=tab, want_costs=True)
    # the oracle's gradient of this very step, before it updates: the entries whose (L2-regularised) gradient lies inside the
    # float32 noise of zero -- noise = the oracle's own distance from the float64 truth, measured above per tensor
    _, g_step, _ = om2.loss_grad(X, Y, B=B, drop=dict(mode=2, p0=0.005, p1=0.005, seed=77, step=0) if drop else None)
    noise = {on: float(np.max(np.abs(np.asarray(rg[on], np.float64).ravel() - np.asarray(g64[on], np.float64).ravel()))) for _, on in pairs}
    near_zero = {on: np.abs(g_step[on].ravel() + float(cfg.l2) * getattr(om2, on).ravel()) <= 8 * noise[on] for _, on in pairs}
    ref_costs = om2.train(X, Y, batch=B, epochs=1, drop_mode=2 if drop else 0, p0=0.005, p1=0.005, seed=77)
    assert abs(c_id[0] - ref_costs[0]) <= LOSS_TOL
    for dn, on in pairs:
        d = np.abs(dm2.get_weights(dn).ravel() - getattr(om2, on).ravel())
        out = d > 1e-5
        # Adam's first update is lr * sign(g): only an entry whose gradient is float32-indistinguishable from zero may differ,
        # and then by at most 2 lr.  Every other entry is held to 1e-5; the outliers are counted, not waved through.
        assert int(np.sum(out & ~near_zero[on])) == 0, (dn, int(np.sum(out & ~near_zero[on])), float(d[out & ~near_zero[on]].max()))
        assert float(out.mean()) <= 2e-3 and np.max(d) <= 2.1e-2, (dn, float(out.mean()), float(np.max(d)))
    # and the next predict sees the updated weights
    assert np.max(np.abs(gm.predict_dataset(dm2, ds, 4096, emb=tab) - om2.predict(X, 4096))) <= 5e-5      # (scores AFTER 20 updates: the drift of COST_TOL_SMALL_SHAPES, test_gpu_ctr.py)


@pytest.mark.parametrize("drop", [False, True])
def test_cfg3_din_full_size_step_vs_oracle(oracle, drop):
    """BASELINE configs[2] at its stated size: DIN cosine, T = 50, D = 16, vocab 26 744, batch 8192 (bench.py default)"""
    one_step_checks(oracle, 0, 52, 50, 16, 53, 26744, 8192, 100, drop)


def test_cfg3_din_euclid_full_size_step_vs_oracle(oracle):
    from goctr_amd import model as gm
    from goctr_amd.recommend import SampleInfo
    U, T, D, Cc, V, B = 52, 50, 16, 53, 26744, 8192
    rng = np.random.default_rng(101)
    emb, ub, it, uf, cf, Y = synth(rng, B, U, T, D, Cc, V)
    om = oracle.CtrModel(0, U, T, D, Cc, att=1)
    om.W0[:] = (rng.standard_normal(om.W0.shape) * 0.15).astype(np.float32)
    om.W1[:] = (rng.standard_normal(om.W1.shape) * 0.15).astype(np.float32)
    om.W2[:] = (rng.standard_normal(om.W2.shape) * 0.15).astype(np.float32)
    dm = gm.DinNet(U, T, D, D, Cc, att=1)
    for n, w in (("mlp0", om.W0), ("mlp1", om.W1), ("mlp2", om.W2), ("att0", om.att0)):
        dm.set_weights(n, w)
    X = oracle.assemble_rows(emb, ub, it, uf, cf)
    si = SampleInfo.from_dims(U, T, D, Cc)
    cost, g, y = gm.loss_grad(dm, si, X, Y, B=B)
    rcost, rg, ry = om.loss_grad(X, Y, B=B)
    c64, g64, y64 = om.loss_grad_f64(X, Y, B=B)
    assert np.max(np.abs(y - ry)) <= LOGIT_TOL and abs(cost - rcost) <= LOSS_TOL
    for dn, on in (("mlp0", "W0"), ("mlp1", "W1"), ("mlp2", "W2"), ("att0", "att0")):
        grad_bound(on, g[dn], rg[on], g64[on])


def test_cfg4_youtube_full_size_step_vs_oracle(oracle):
    """BASELINE configs[3] per-GPU slice at its stated size: YouTube-DNN, vocab 10^7 x 64-d (2.56 GB table), batch 16384"""
    one_step_checks(oracle, 1, 52, 50, 64, 53, 10_000_000, 16384, 102, True)


def test_cfg2_mlp_full_size_step_vs_oracle(oracle):
    """BASELINE configs[1] at its stated size: sklearn-port MLP [281,100,1] relu/adam, batch 4096, float64: loss and
    packed gradients of one full batch at 1e-9 relative, then 4 updates (per-parameter Adam, quirk Q7) vs the oracle"""
    from goctr_amd import mlp as gmlp
    rng = np.random.default_rng(103)
    units, B = [281, 100, 1], 4096
    X = rng.random((4 * B, 281), dtype=np.float32)
    Y = (X[:, 0] + X[:, 5] * X[:, 9] > 1.0).astype(np.float32).reshape(-1, 1)
    clf = gmlp.MLPClassifier([100], "relu", "adam", 1e-5)
    clf.BatchSize, clf.MaxIter, clf.Tol, clf.Shuffle = B, 1, -1.0, False
    theta0 = clf.init_params(units, rng)
    clf.create(units, B, theta0.copy())
    loss, g = clf.loss_grad(X[:B].astype(np.float64), Y[:B].astype(np.float64))
    cfg = oracle.mlp_cfg(units, "relu", alpha=1e-5)
    rloss, rg = oracle.mlp_loss_grad(cfg, theta0.copy(), X[:B].astype(np.float64), Y[:B].astype(np.float64))
    assert loss == pytest.approx(rloss, rel=1e-9)
    assert np.max(np.abs(g - rg)) <= 1e-9 * np.max(np.abs(rg)) + 1e-15
    clf.Fit(X, Y, theta0=theta0.copy(), perm=np.arange(4 * B, dtype=np.int32)[None, :])
    theta = theta0.copy()
    ref = oracle.mlp_fit(cfg, theta, oracle.MlpOptimizer("adam", theta.size), X.astype(np.float64), Y.astype(np.float64), B, 1,
                         tol=-1.0, perm=np.arange(4 * B, dtype=np.int32)[None, :])
    assert clf.LossCurve[0] == pytest.approx(ref[0], rel=1e-8)
    assert np.max(np.abs(clf.get_params() - theta)) <= 1e-7 * np.max(np.abs(theta)) + 1e-10


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] at its stated size: item2vec SkipGram + hierarchical softmax, 10^7-word corpus, window 5.
# Hogwild is racy by design -- in the reference too (goroutines over shared matrices) -- so parity can only be
# statistical: the run at the parallelism bench.py uses (32 768 workers) must reach the HS loss and the neighbour
# structure of the oracle's 16-thread Hogwild (the reference's own scheme: runtime.NumCPU() slices) on the same corpus
# from the same initial vectors.  Windows are clipped at the 16 slice ends on both sides (quirk Q18).

def _session_corpus(rng, V, n, topics=64, mean_len=70):
    """user sessions: every session has a topic; 80 % of its tokens are Zipf draws from the topic's
    items (item % topics == topic), 20 % Zipf draws from the whole vocabulary"""
    n_sess = n // mean_len + 1
    lens = rng.poisson(mean_len, size=n_sess).clip(5)
    topic = np.repeat(rng.integers(0, topics, size=n_sess), lens)[:n]
    per = V // topics
    zr = (rng.zipf(1.2, size=n) - 1)
    in_topic = rng.random(n) < 0.8
    tok = np.where(in_topic, (zr % per) * topics + topic, zr % V)
    return tok.astype(np.int32), topics


def _hs_loss(param, aux, paths, pairs):
    """mean negative log-likelihood of the Huffman codes of `center` given the vector of `context` (the quantity
    hierarchicalSoftmax.optim ascends, optimizer.go:107-129), exact sigmoid, float64"""
    off, nodes, codes = paths
    tot, cnt = 0.0, 0
    for center, ctx in pairs:
        v = param[ctx]
        nd = nodes[off[center]:off[center + 1]]
        cd = codes[off[center]:off[center + 1]].astype(np.float64)
        x = aux[nd] @ v
        # label = 1 - code: log sigma(x) for code 0, log(1 - sigma(x)) for code 1
        tot += np.sum(np.logaddexp(0.0, -x) * (1 - cd) + np.logaddexp(0.0, x) * cd)
        cnt += nd.size
    return tot / cnt


def _neighbours(P, words, k=10):
    Pn = P / np.maximum(np.linalg.norm(P, axis=1, keepdims=True), 1e-30)
    S = Pn[words] @ Pn.T
    S[np.arange(len(words)), words] = -2.0
    return np.argsort(-S, axis=1)[:, :k]


def test_cfg5_item2vec_full_size_hogwild_vs_oracle(oracle):
    from goctr_amd import embedding as ge
    rng = np.random.default_rng(105)
    V, dim, n, streams, slices = 10681, 16, 10_000_000, 32768, 16
    doc, topics = _session_corpus(rng, V, n)
    counts = np.bincount(doc, minlength=V) + 1
    p0 = (rng.random((V, dim)) - 0.5) / dim                       # word2vec.go:103-111 init
    m = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=False, streams=streams, slices=slices)
    m.create(counts, p0.copy())
    m.train_pass(doc, n, None, lr=0.025)
    gp, ga = m.get_param(), m.get_aux()
    assert np.all(np.isfinite(gp)) and np.all(np.isfinite(ga))

    oracle.set_threads(slices)
    cfg = oracle.w2v_cfg(dim=dim, optimizer="hs")
    paths = oracle.huffman_paths(counts)
    op, oa = p0.copy(), np.zeros((V - 1, dim))
    oracle.w2v_train_hogwild(cfg, doc, slices, None, op, oa, paths, oracle.sigmoid_table(), 0.025, n)

    # held-out (center, context) pairs from the corpus itself (adjacent tokens of random positions)
    pos = rng.integers(1, n - 1, size=4000)
    pairs = list(zip(doc[pos].tolist(), doc[pos + 1].tolist()))
    l0 = _hs_loss(p0, np.zeros((V - 1, dim)), paths, pairs)     # = ln 2 per node before training
    lg, lo = _hs_loss(gp, ga, paths, pairs), _hs_loss(op, oa, paths, pairs)
    print(f"HS loss per node: init {l0:.4f}  device {lg:.4f}  oracle(16 threads) {lo:.4f}")
    assert abs(l0 - np.log(2.0)) < 1e-9
    assert lo < 0.9 * l0 and lg < 0.9 * l0                       # both learned
    assert abs(lg - lo) <= 0.03 * lo                             # ... to the same loss

    # neighbour structure of the 300 most frequent items: same-topic share of the top-10, and the two runs' overlap
    words = np.argsort(-counts)[:300]
    ng, no = _neighbours(gp, words), _neighbours(op, words)
    purity = lambda nb: float(np.mean((nb % topics) == (words % topics)[:, None]))
    overlap = float(np.mean([len(set(a) & set(b)) / 10.0 for a, b in zip(ng, no)]))
    print(f"top-10 same-topic share: device {purity(ng):.3f} oracle {purity(no):.3f} (chance {1 / topics:.3f}); overlap {overlap:.3f}")
    assert purity(no) > 10.0 / topics and purity(ng) >= 0.9 * purity(no)
    # (a topic holds V / topics = 166 items: two equally good racy runs rank them differently, so the lists overlap
    #  far less than they are pure -- but far more than two random lists of the vocabulary, 10 / V = 0.001)
    assert overlap >= 50 * 10.0 / V


def test_item2vec_stress_point_v1e6_d64_hogwild_vs_oracle(oracle):
    """cfg5's stress point: vocabulary 10^6, 64-d vectors -- 512 MB of float64 parameters + as many node
    vectors, Huffman paths of ~22 nodes -- where the LDS-cached hot rows are a much smaller share of the traffic than at
    V = 10 681 / D = 16.  10^6 words (the oracle's 16-thread Hogwild on the box's CPU quota is what bounds the size): same
    statistical gate as above -- HS loss per path node within 3 % of the oracle's run from the same initial vectors"""
    from goctr_amd import embedding as ge
    rng = np.random.default_rng(106)
    V, dim, n, streams, slices = 1_000_000, 64, 1_000_000, 32768, 16
    doc, topics = _session_corpus(rng, V, n, topics=64, mean_len=70)
    counts = np.bincount(doc, minlength=V) + 1
    p0 = ((rng.random((V, dim), dtype=np.float32) - 0.5) / dim).astype(np.float64)
    m = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=False, streams=streams, slices=slices)
    m.create(counts, p0.copy())
    m.train_pass(doc, n, None, lr=0.025)
    gp, ga = m.get_param(), m.get_aux()
    assert np.all(np.isfinite(gp)) and np.all(np.isfinite(ga))
    oracle.set_threads(slices)
    cfg = oracle.w2v_cfg(dim=dim, optimizer="hs")
    paths = oracle.huffman_paths(counts)
    assert _same_paths(m.get_paths(), paths)                     # the 10^6-word tree: product builder == oracle builder
    op, oa = p0.copy(), np.zeros((V - 1, dim))
    oracle.w2v_train_hogwild(cfg, doc, slices, None, op, oa, paths, oracle.sigmoid_table(), 0.025, n)
    pos = rng.integers(1, n - 1, size=3000)
    pairs = list(zip(doc[pos].tolist(), doc[pos + 1].tolist()))
    l0 = _hs_loss(p0, np.zeros((V - 1, dim)), paths, pairs)
    lg, lo = _hs_loss(gp, ga, paths, pairs), _hs_loss(op, oa, paths, pairs)
    print(f"V=1e6 D=64: HS loss per node: init {l0:.4f}  device {lg:.4f}  oracle(16 threads) {lo:.4f}")
    assert abs(l0 - np.log(2.0)) < 1e-9
    assert lo < 0.97 * l0 and lg < 0.97 * l0
    # not worse than the oracle's Hogwild by more than 3 %; in this short-corpus regime the device run (whose hottest rows are
    # averaged over the workgroups: lower-variance steps on the top of the tree) lands BELOW the oracle's loss (0.595 vs 0.624)
    assert lg <= 1.03 * lo and lg >= 0.85 * lo


def _same_paths(a, b):
    return all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a, b))


W2V_W8_GATE = r'''
from goctr_amd import embedding as ge
d = np.load(%(inp)r)
doc, counts, p0 = d["doc"], d["counts"], d["p0"]
res = {}
for name, every in (("periodic", 0), ("once", -1)):
    capi.engine_select(0)
    m = ge.Word2Vec(dim=16, optimizer="hs", deterministic=False, streams=32768, slices=16, devices=W, exchange_every=every)
    m.create(counts, p0.copy())
    m.train_pass(doc, doc.size, None, lr=0.025)
    res[name + "_p"] = m.get_param(); res[name + "_a"] = m.get_aux()
    # the replicas hold the same bits after the last exchange
    m2p = m.get_param()
    assert np.array_equal(m2p, res[name + "_p"])
np.savez(%(out)r, **res)
'''


def test_cfg5_item2vec_w8_exchange_cadence_vs_oracle(oracle, tmp_path):
    """BASELINE configs[4] across devices: the 10^7-word corpus on W = 8 logical ranks
    (loop-back communicator on one GPU: every W > 1 code path), ONE goctr_w2v_train call with cfg.devices = 8, `iter 1`.
    Two cadences of the parameter-delta exchange (per row: the average over the ranks that updated it, csrc/w2v.hip
    exchange_deltas): every update_lr_batch = 10^5 words per rank (the default: 13 segments per pass, aligned with the
    reference's observer, word2vec.go:223-233 / options.go:55) and once per pass.  Gate: HS loss per path node within 3 % of the
    oracle's 16-thread Hogwild run from the same initial vectors -- REQUIRED of the default cadence, measured and reported for
    once-per-pass (the oracle-kernel simulation scripts/w2v_dp_sim.py predicts 0.564 vs 0.626 against 0.559).  Round 5's first
    run of this test found what rounds 3-4 shipped unmeasured: the plain SUM of the eight deltas diverges (loss 6.2 once per
    pass, 8.9 at 13 exchanges; profiles/r05_w2v_dp_exchange.txt).  Both numbers are printed and written to gpurun_out/."""
    import json
    from test_gpu_multi import run_script
    rng = np.random.default_rng(105)
    V, dim, n, slices = 10681, 16, 10_000_000, 16
    doc, topics = _session_corpus(rng, V, n)
    counts = np.bincount(doc, minlength=V) + 1
    p0 = (rng.random((V, dim)) - 0.5) / dim
    inp = str(tmp_path / "w2v_gate_in.npz")
    np.savez(inp, doc=doc, counts=counts, p0=p0)
    r = run_script(W2V_W8_GATE, tmp_path, "w2v_w8_gate", timeout=1200, W=8, inp=inp)
    oracle.set_threads(slices)
    cfg = oracle.w2v_cfg(dim=dim, optimizer="hs")
    paths = oracle.huffman_paths(counts)
    op, oa = p0.copy(), np.zeros((V - 1, dim))
    oracle.w2v_train_hogwild(cfg, doc, slices, None, op, oa, paths, oracle.sigmoid_table(), 0.025, n)
    oracle.set_threads(1)
    pos = rng.integers(1, n - 1, size=4000)
    pairs = list(zip(doc[pos].tolist(), doc[pos + 1].tolist()))
    l0 = _hs_loss(p0, np.zeros((V - 1, dim)), paths, pairs)
    lo = _hs_loss(op, oa, paths, pairs)
    lp = _hs_loss(r["periodic_p"], r["periodic_a"], paths, pairs)
    l1 = _hs_loss(r["once_p"], r["once_a"], paths, pairs)
    line = {"workload": "item2vec cfg5, 10^7 words, W = 8 (loop-back)", "hs_loss_init": l0, "hs_loss_oracle_16_threads": lo,
            "hs_loss_exchange_every_1e5_words": lp, "hs_loss_exchange_once_per_pass": l1,
            "rel_periodic": lp / lo - 1, "rel_once": l1 / lo - 1}
    print(json.dumps(line))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(line, open(os.path.join(out_dir, "w2v_w8_gate.json"), "w"))
    assert np.all(np.isfinite(r["periodic_p"])) and np.all(np.isfinite(r["once_p"]))
    assert lo < 0.9 * l0 and lp < 0.9 * l0
    assert abs(lp - lo) <= 0.03 * lo, line                       # the default cadence holds the gate
    # once per pass is reported, not required to hold the gate (eight replicas averaged after training apart for a whole pass)
    assert l1 < l0, line
