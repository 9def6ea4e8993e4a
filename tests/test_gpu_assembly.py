"""GPU parity for the device-side sample assembly: behaviour-cache lookups and the key -> row
gather must equal the oracle's restatement of cache.go:71-94 / rcmd.go:460-536 BIT FOR BIT (index / gather work)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kats.json")))


def make_cache(rng, n_users, V, max_len):
    from goctr_amd import ubcache
    ubc = ubcache.NewUserBehaviorCache()
    seqs = {}
    for u in range(n_users):
        n = int(rng.integers(0, max_len + 1)) if u % 7 else 0           # some users with an empty history
        ts = np.sort(rng.integers(1, 1000, size=n))[::-1]               # descending, with duplicates
        items = rng.integers(0, V, size=n)
        seqs[u] = (ts.astype(np.int64), items.astype(np.int32))
        ubc.Set(u, ubcache.TimeSeq(ts.tolist(), items.tolist()))
    return ubc, seqs


def test_reference_kats():
    from goctr_amd import ubcache
    k = KATS["ubcache_filter"]
    ubc = ubcache.NewUserBehaviorCache()
    ubc.Set(1, ubcache.TimeSeq(k["ts"], k["items"]))
    for c in k["cases"]:
        got = ubc.Get(1, c["max_ts"], c["max_len"])
        assert got.Items == c["expect"] and got.Ts == c["expect"]       # (Items == Ts in the reference's fixture)
    with pytest.raises(KeyError):
        ubc.Get(3, 0, 0)
    ubc.Delete(1)
    with pytest.raises(KeyError):
        ubc.Get(1, 0, 0)


@pytest.mark.parametrize("T", [1, 7, 50, 64, 130])
def test_batch_lookup_bit_exact(oracle, T):
    rng = np.random.default_rng(T)
    ubc, seqs = make_cache(rng, 60, 500, 200)
    users = rng.integers(0, 60, size=3000)
    ts = rng.integers(0, 1100, size=3000)
    ts[rng.random(3000) < 0.1] = 0                                      # maxTs == 0: from the newest entry
    got = ubc.get_batch(users, ts, T)
    for r in range(users.size):
        s_ts, s_it = seqs[int(users[r])]
        ref = oracle.ubcache_filter(s_ts, s_it, int(ts[r]), T)
        exp = np.full(T, -1, np.int32)
        exp[:ref.size] = ref
        assert np.array_equal(got[r], exp), (r, users[r], ts[r])


def test_dataset_from_keys_equals_dataset_from_ids(oracle):
    from goctr_amd import capi, model as gm
    rng = np.random.default_rng(11)
    n_users, n_items, U, Cc, T, D, rows = 40, 300, 52, 53, 50, 16, 2048
    ubc, seqs = make_cache(rng, n_users, n_items, 120)
    user_table = rng.random((n_users, U), dtype=np.float32)
    item_table = rng.random((n_items, Cc), dtype=np.float32)
    users = rng.integers(0, n_users, size=rows).astype(np.int32)
    items = rng.integers(0, n_items, size=rows).astype(np.int32)
    ts = rng.integers(1, 1100, size=rows).astype(np.int64)
    y = (rng.random(rows) < 0.5).astype(np.float32)
    ds_k = gm.Dataset.keys(ubc, user_table, item_table, users, items, ts, y, T)
    # oracle: the same assembly on the CPU
    off = np.zeros(n_users + 1, np.int64)
    for u in range(n_users):
        off[u + 1] = off[u] + seqs[u][0].size
    seq_items = np.concatenate([seqs[u][1] for u in range(n_users)])
    seq_ts = np.concatenate([seqs[u][0] for u in range(n_users)])
    ub, uf, cf = oracle.assemble_keys(off, seq_items, seq_ts, user_table, item_table, users, items, ts, T)
    gub, guf, gcf = ds_k.get_ids()
    assert np.array_equal(gub, ub) and np.array_equal(guf, uf) and np.array_equal(gcf, cf)      # bit-exact
    # and the model sees the same thing either way: identical training trajectory
    emb = (rng.standard_normal((n_items, D)) * 0.25).astype(np.float32)
    tab = gm.EmbeddingTable(emb)
    ds_i = gm.Dataset.ids(ub, items, uf, cf, y)
    w = []
    for ds in (ds_k, ds_i):
        m = gm.DinNet(U, T, D, D, Cc).init_gaussian(np.random.default_rng(1))
        cfg = capi.default_train_cfg(batch=512, epochs=1, dropout_mode=0)
        gm.train_steps(m, ds, cfg, 4, emb=tab)
        capi.sync()
        w.append(np.concatenate([m.get_weights(n).ravel() for n in ("mlp0", "mlp1", "mlp2", "att0")]))
    assert np.array_equal(w[0], w[1])
