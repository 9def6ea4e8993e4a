"""GPU parity for recommend.BatchPredict / Rank (recommend/rcmd.go:248-337): sample keys -> device-side row
assembly -> predict -> scores through goctr_batch_predict / goctr_rank, against the oracle's composition of the same
steps (rows bit-exact by construction of the assembly kernels, scores <= 1e-5), including the reference's error
behaviour: failing keys score as the all-zero row, a failing first key aborts, a failing last key returns y AND an error."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def build(oracle, rng, kind, n_users=40, n_items=300, n_emb_only=20, T=10, D=16, U=7, Cc=9):
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
        n = int(rng.integers(0, 30))
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
    assert np.max(np.abs(costs - ref)) <= 5e-5
    s = gr.Rank(model, uids[0], iids[:20], now=600)
    keys = [gr.Sample(uids[0], i, 0.0, 600) for i in iids[:20]]
    ref_s, _ = oracle_scores(oracle, rs, om, keys, model.PredBatchSize)
    assert np.max(np.abs(np.array([x.Score for x in s], np.float32) - ref_s)) <= 1e-4
