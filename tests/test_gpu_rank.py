"""GPU parity for recommend.BatchPredict / Rank (recommend/rcmd.go:248-337): sample keys -> device-side row
assembly -> predict -> scores through goctr_batch_predict / goctr_rank, against the oracle's composition of the same
steps (rows bit-exact by construction of the assembly kernels, scores <= 1e-5), including the reference's error
behaviour: failing keys score as the all-zero row, a failing first key aborts, a failing last key returns y AND an error."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def build(oracle, rng, kind, n_users=40, n_items=300, n_emb_only=20, T=10, D=16, U=7, Cc=9, max_hist=30):
    from goctr_amd import model as gm, recommend as gr, ubcache
    # ids are arbitrary ints like in the reference (not dense): users 1000+3k, items 7+5k
    uids = [1000 + 3 * k for k in range(n_users)]
    iids = [7 + 5 * k for k in range(n_items)]
    extra = [10_000 + k for k in range(n_emb_only)]                 # items with an embedding but no feature row
    ufeat = {u: rng.random(U, dtype=np.float32) for u in uids}
    ifeat = {i: rng.random(Cc, dtype=np.float32) for i in iids}
    iemb = {i: (rng.standard_normal(D) * 0.3).astype(np.float32) for i in iids[: n_items - 15] + extra}   # 15 items lack one
    ubc = ubcache.NewUserBehaviorCache()
    for u in uids:
        n = int(rng.integers(0, max_hist))
        ts = np.sort(rng.integers(1, 1000, size=n))[::-1]
        its = rng.choice(iids + extra + [999_999], size=n)           # incl. an item unknown to every table
        ubc.Set(u, ubcache.TimeSeq(ts.tolist(), [int(x) for x in its]))
    rs = gr.DeviceRecSys(ufeat, ifeat, iemb, ubc, T=T)
    om = oracle.CtrModel(kind, U, T, D, Cc)
    om.W0[:] = (rng.standard_normal(om.W0.shape) * 0.2).astype(np.float32)
    om.W1[:] = (rng.standard_normal(om.W1.shape) * 0.2).astype(np.float32)
    om.W2[:] = (rng.standard_normal(om.W2.shape) * 0.2).astype(np.float32)
    if kind == 0:
        om.att0[:] = (1 + 0.3 * rng.standard_normal(T)).astype(np.float32)
    net = (gm.DinNet if kind == 0 else gm.YoutubeDnn)(U, T, D, D, Cc)
    for n, w in (("mlp0", om.W0), ("mlp1", om.W1), ("mlp2", om.W2)):
        net.set_weights(n, w)
    if kind == 0:
        net.set_weights("att0", om.att0)
    return rs, om, net, uids, iids, extra


def oracle_scores(oracle, rs, om, keys, batch):
    """the reference's BatchPredict with the oracle's pieces: keys -> dense indices -> rows -> predict"""
    users, items, ts = rs.keys(keys)
    dc = rs._dense_cache
    ids = sorted(dc.ub)
    off = np.zeros(len(ids) + 1, np.int64)
    for k, u in enumerate(ids):
        off[k + 1] = off[k] + len(dc.ub[u].Ts)
    seq_items = np.concatenate([np.asarray(dc.ub[u].Items, np.int32) for u in ids])
    seq_ts = np.concatenate([np.asarray(dc.ub[u].Ts, np.int64) for u in ids])
    X, failed = oracle.batch_predict_rows(rs.emb.get_rows(), off, seq_items, seq_ts, rs.user_table, rs.item_table, users, items,
                                          ts, rs.T)
    return om.predict(X, batch), failed


@pytest.mark.parametrize("kind", [0, 1])
def test_batch_predict_matches_oracle(oracle, kind):
    from goctr_amd import recommend as gr
    rng = np.random.default_rng(20 + kind)
    rs, om, net, uids, iids, extra = build(oracle, rng, kind)
    n = 1237
    keys = [gr.Sample(int(rng.choice(uids)), int(rng.choice(iids)), 0.0, int(rng.integers(0, 1100))) for _ in range(n)]
    for k in (5, 77, 400):
        keys[k] = gr.Sample(4242, keys[k].ItemId, 0.0, 50)           # unknown user  -> GetUserFeature error -> zero row
    for k in (9, 78, 1000):
        keys[k] = gr.Sample(keys[k].UserId, extra[k % len(extra)], 0.0, 50)   # embedding-only item -> GetItemFeature error
    keys[100] = gr.Sample(keys[100].UserId, keys[100].ItemId, 0.0, 0)  # maxTs == 0: from the newest (cache.go:72-74)
    model = gr.Predictor(rs, net, predBatchSize=256)                  # 1237 rows @ 256: padded last batch
    y = gr.BatchPredict(model, keys)
    ref, failed = oracle_scores(oracle, rs, om, keys, 256)
    assert y.shape == (n, 1)
    assert failed.sum() == 6
    assert np.max(np.abs(y[:, 0] - ref)) <= 1e-5
    # failing keys score exactly like an all-zero row
    zero = om.predict(np.zeros((1, om.xcols), np.float32), 1)[0]
    assert np.all(np.abs(y[failed.astype(bool), 0] - zero) <= 1e-5)


@pytest.mark.parametrize("kind,D,T,max_hist", [(0, 16, 10, 30), (0, 8, 70, 400), (1, 64, 20, 400), (0, 4, 5, 30), (0, 12, 10, 30)])
def test_key_lookup_inside_the_attention_kernel_equals_assembled_rows(oracle, monkeypatch, kind, D, T, max_hist):
    """serving passes run as one launch (ctr_serve16_kernel: key lookup + attention + forward chain; D = 8, 16, 64), look the
    keys up inside attn_fwd (attn_fwd_keys_kernel; embedding widths 4 .. 64, powers of two; GOCTR_SERVE_ONE_LAUNCH=0) or
    assemble the rows first (assemble_keys_kernel: other widths, GOCTR_SERVE_FUSE=0): same scores bit for bit, all against
    the oracle; histories longer than 256 entries take the bisection instead of the ballot search, T > 64 two id blocks"""
    from goctr_amd import recommend as gr
    rng = np.random.default_rng(500 + D + T)
    rs, om, net, uids, iids, extra = build(oracle, rng, kind, T=T, D=D, max_hist=max_hist)
    n = 700
    keys = [gr.Sample(int(rng.choice(uids)), int(rng.choice(iids)), 0.0, int(rng.integers(0, 1100))) for _ in range(n)]
    keys[3] = gr.Sample(4242, keys[3].ItemId, 0.0, 50)
    keys[10] = gr.Sample(keys[10].UserId, extra[0], 0.0, 50)
    keys[11] = gr.Sample(keys[11].UserId, keys[11].ItemId, 0.0, 0)
    model = gr.Predictor(rs, net, predBatchSize=256)
    ys = []
    # one launch per pass (ctr_serve16_kernel: D = 8, 16, 64) / key lookup inside attn_fwd + the forward chain / assembled rows first
    for fuse, one in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("GOCTR_SERVE_FUSE", fuse)
        monkeypatch.setenv("GOCTR_SERVE_ONE_LAUNCH", one)
        ys.append(gr.BatchPredict(model, keys)[:, 0].copy())
    assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2])
    # the one-launch pass again with the host waiting on the stream for every pass / watching the workgroups' stamps in the
    # pinned buffer whatever the pass size (default: passes of up to 512 rows -- the 256-row passes above)
    monkeypatch.setenv("GOCTR_SERVE_FUSE", "1")
    monkeypatch.setenv("GOCTR_SERVE_ONE_LAUNCH", "1")
    for rows in ("0", "100000"):
        monkeypatch.setenv("GOCTR_SERVE_POLL_ROWS", rows)
        for _ in range(3):                             # (stamps are pass numbers: consecutive passes must not see old ones)
            assert np.array_equal(gr.BatchPredict(model, keys)[:, 0], ys[0]), rows
    monkeypatch.delenv("GOCTR_SERVE_POLL_ROWS")
    # ... and with the kernels reading the keys from the pinned host buffer instead of from device memory the host stored
    # them into over the PCIe BAR (the default where the system has a large BAR)
    monkeypatch.setenv("GOCTR_SERVE_BAR", "0")
    assert np.array_equal(gr.BatchPredict(model, keys)[:, 0], ys[0])
    monkeypatch.delenv("GOCTR_SERVE_BAR")
    ref, failed = oracle_scores(oracle, rs, om, keys, 256)
    assert failed.sum() == 2
    assert np.max(np.abs(ys[0] - ref)) <= 1e-5


def test_rank_and_error_behaviour(oracle):
    from goctr_amd import recommend as gr
    rng = np.random.default_rng(31)
    rs, om, net, uids, iids, extra = build(oracle, rng, 0)
    model = gr.Predictor(rs, net, predBatchSize=64)
    cand = [int(x) for x in rng.choice(iids, size=150, replace=False)]
    cand[40] = 123_456_789                                           # unknown item in the middle: zero row, no error
    scores = gr.Rank(model, uids[3], cand, now=500)
    assert [s.ItemId for s in scores] == cand
    ref, failed = oracle_scores(oracle, rs, om, [gr.Sample(uids[3], i, 0.0, 500) for i in cand], 64)
    assert failed.sum() == 1 and failed[40]
    assert np.max(np.abs(np.array([s.Score for s in scores], np.float32) - ref)) <= 1e-5
    # first key fails: BatchPredict returns the error (rcmd.go:293-296)
    with pytest.raises(gr.SampleVectorError):
        gr.Rank(model, 4242, cand, now=500)
    # last key fails: y AND err come back (named result never cleared, rcmd.go:291); Rank drops the scores (:258-260)
    with pytest.raises(gr.SampleVectorError) as ei:
        gr.BatchPredict(model, [gr.Sample(uids[3], i, 0.0, 500) for i in cand[:10] + [123_456_789]])
    assert ei.value.y.shape == (11, 1)
    assert np.max(np.abs(ei.value.y[:10, 0] - ref[:10])) <= 1e-5
    with pytest.raises(gr.SampleVectorError):
        gr.Rank(model, uids[3], cand[:10] + [123_456_789], now=500)
    # empty candidate list
    assert gr.Rank(model, uids[3], [], now=1) == []


def test_train_from_keys_then_rank(oracle):
    """recommend.Train (rcmd.go:187-246) end to end on the device: GetSample drops keys without features (:379-382),
    model.Train runs on the assembled id-mode rows, the Predictor ranks."""
    from goctr_amd import model as gm, recommend as gr
    rng = np.random.default_rng(41)
    rs, om, net, uids, iids, extra = build(oracle, rng, 0)
    n = 600
    samples = [gr.Sample(int(rng.choice(uids)), int(rng.choice(iids)), float(rng.random() < 0.5), int(rng.integers(1, 1000)))
               for _ in range(n)]
    samples[7] = gr.Sample(4242, samples[7].ItemId, 1.0, 5)          # dropped
    ds, si, kept = gr.GetSample(rs, samples)
    assert kept.size == n - 1 and 7 not in kept
    # the assembled rows are the oracle's rows, bit for bit
    users, items, ts = rs.keys([samples[i] for i in kept])
    ub, uf, cf = ds.get_ids()
    X = oracle.assemble_rows(rs.emb.get_rows(), ub, items, uf, cf)
    Y = np.array([samples[i].Label for i in kept], np.float32)
    ref = om.train(X, Y, batch=100, epochs=3)
    model, costs = gr.Train(rs, samples, net, batchSize=100, epochs=3, earlyStop=0, dropout_seed=None)
    assert np.max(np.abs(costs - ref)) <= 5e-5          # (reduced-size shape: see COST_TOL_SMALL_SHAPES in test_gpu_ctr.py)
    s = gr.Rank(model, uids[0], iids[:20], now=600)
    keys = [gr.Sample(uids[0], i, 0.0, 600) for i in iids[:20]]
    ref_s, _ = oracle_scores(oracle, rs, om, keys, model.PredBatchSize)
    assert np.max(np.abs(np.array([x.Score for x in s], np.float32) - ref_s)) <= 1e-4


# ---------------------------------------------------------------------------------------------------------------------
# Serving as the reference calls it (recommend/api.go:106-131): many handler goroutines, each one Rank call with a short
# candidate list.  goctr_rank / goctr_batch_predict run on serving slots (own stream, pinned staging) under a shared lock
# of the model, and small calls that arrive while a pass is in flight are coalesced into one pass.  None of that may
# change a score: every concurrent answer must be bit-equal to the single-threaded answer of the same request.

def _rank_raw(net, rs, user_idx, item_idx, now):
    import ctypes as C
    from goctr_amd import capi
    items = np.ascontiguousarray(item_idx, np.int32)
    y = np.empty(items.size, np.float32)
    failed = np.zeros(items.size, np.uint8)
    nf = C.c_int64(0)
    capi.check(capi.load().goctr_rank(net._h, rs._h, C.c_int32(int(user_idx)), capi.ptr(items, C.c_int32), C.c_int64(items.size),
                                      C.c_int64(now), C.c_int(4096), capi.ptr(y, C.c_float), capi.ptr(failed, C.c_uint8), C.byref(nf)))
    return y, failed, nf.value


@pytest.mark.parametrize("coalesce", ["1024", "0"])
def test_concurrent_rank_calls_are_bit_equal_to_sequential(oracle, coalesce, monkeypatch):
    import threading
    monkeypatch.setenv("GOCTR_SERVE_COALESCE", coalesce)
    rng = np.random.default_rng(51)
    rs, om, net, uids, iids, extra = build(oracle, rng, 0, n_users=64, n_items=2000, T=50, U=52, Cc=53)
    n_items_tab = rs.item_table.shape[0]
    reqs = []
    for q in range(48):
        n = int(rng.choice([1, 7, 32, 100, 256, 700]))
        items = rng.integers(0, n_items_tab, size=n).astype(np.int32)
        if q % 5 == 0 and n > 3:
            items[n // 2] = n_items_tab + 17                         # a key without features in the middle: zero row + flag
        reqs.append((int(rng.integers(0, len(uids))), items, int(rng.integers(1, 1100))))
    want = [_rank_raw(net, rs, *r) for r in reqs]
    errs, got = [], {}

    def worker(t):
        try:
            for rep in range(6):
                for q in range(t % 4, len(reqs), 4):                   # (threads t and t + 4 race on identical requests)
                    got[(t, rep, q)] = _rank_raw(net, rs, *reqs[q])
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert len(got) == 8 * 6 * 12
    for (t, rep, q), (y, failed, nf) in got.items():
        assert np.array_equal(y, want[q][0]) and np.array_equal(failed, want[q][1]) and nf == want[q][2], (t, rep, q)
    assert any(w[2] > 0 for w in want)


def test_serving_while_training_the_same_model(oracle):
    """a training call on a model waits for the serving passes in flight and they for it (shared / exclusive model lock,
    and the slot streams wait for the main stream's last weight write): scores returned after training equal a fresh call's,
    and training lands on the bits of an undisturbed run"""
    import threading
    from goctr_amd import capi, model as gm
    rng = np.random.default_rng(61)
    rs, om, net, uids, iids, extra = build(oracle, rng, 0, n_users=64, n_items=2000, T=50, U=52, Cc=53)
    twin = gm.DinNet(52, 50, 16, 16, 53)
    for n in ("mlp0", "mlp1", "mlp2", "att0"):
        twin.set_weights(n, net.get_weights(n))
    U, T, D, Cc, V, rows, B = 52, 50, 16, 53, rs.emb.get_rows().shape[0], 4096, 512
    ub = rng.integers(-1, V, size=(rows, T)).astype(np.int32)
    it = rng.integers(0, V, size=rows).astype(np.int32)
    uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
    Y = (uf[:, 0] > 0.5).astype(np.float32)
    ds = gm.Dataset.ids(ub, it, uf, cf, Y)
    cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2, p0=0.005, p1=0.005, seed=9)
    want_costs = np.concatenate([gm.train_steps(twin, ds, cfg, 8, first_batch=8 * r, emb=rs.emb, want_costs=True) for r in range(5)])
    items = rng.integers(0, rs.item_table.shape[0], size=200).astype(np.int32)
    stop, errs, n_calls = threading.Event(), [], [0]

    def server():
        try:
            while not stop.is_set():
                y, _, _ = _rank_raw(net, rs, 5, items, 700)
                assert np.all(np.isfinite(y))
                n_calls[0] += 1
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=server) for _ in range(3)]
    for t in th:
        t.start()
    costs = np.concatenate([gm.train_steps(net, ds, cfg, 8, first_batch=8 * r, emb=rs.emb, want_costs=True) for r in range(5)])
    stop.set()
    for t in th:
        t.join()
    assert not errs, errs
    assert n_calls[0] > 0
    assert np.array_equal(costs, want_costs)
    assert np.array_equal(net.get_weights("mlp0"), twin.get_weights("mlp0"))
    y_after, _, _ = _rank_raw(net, rs, 5, items, 700)
    y_twin, _, _ = _rank_raw(twin, rs, 5, items, 700)
    assert np.array_equal(y_after, y_twin)


def test_rank_bench_binary_runs_and_is_bit_equal():
    """goctr_amd/host/rank_bench (std::thread callers above the C-ABI, what bench.py's rank_* fields come from): exits 0 only
    if every concurrent answer was bit-equal to the single-threaded one"""
    import json
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "goctr_amd", "host", "rank_bench")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    r = subprocess.run([exe, "--threads", "1,4", "--n", "16,300", "--seconds", "0.15"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["bit_equal_to_single_threaded"] is True and len(d["results"]) == 8
    assert all(e["calls"] > 0 and e["mismatched_calls"] == 0 for e in d["results"])
