"""GPU: trainable-embedding EXTENSION (csrc/emb_train.h) against its float64 oracle (oracle/orc_embtrain.c, checked
by finite differences in tests/test_oracle_embtrain.py).  go-ctr itself keeps the embeddings frozen, so
this is parity with the build's own restatement, tolerance 1e-4 of the largest row update (float32 forward/backward
vs float64), plus exact properties: untouched rows keep their bits, the update is bit-reproducible, lr = 0 is the
reference's frozen behaviour."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(oracle, kind, att, U, T, D, Cc, V, rows, seed):
    from goctr_amd import model as gm
    rng = np.random.default_rng(seed)
    okind = oracle.DIN if kind == "din" else oracle.YOUTUBE
    om = oracle.CtrModel(okind, U, T, D, Cc, att=att)
    om.W0[:] = rng.standard_normal(om.W0.shape) * 0.2
    om.W1[:] = rng.standard_normal(om.W1.shape) * 0.2
    om.W2[:] = rng.standard_normal(om.W2.shape) * 0.3
    om.att0[:] = 1.0 + 0.3 * rng.standard_normal(T)
    cls = gm.DinNet if kind == "din" else gm.YoutubeDnn
    m = cls(U, T, D, D, Cc, att=att)
    for n, w in (("mlp0", om.W0), ("mlp1", om.W1), ("mlp2", om.W2)):
        m.set_weights(n, w)
    if kind == "din":
        m.set_weights("att0", om.att0)
    E = (rng.standard_normal((V, D)) * 0.5).astype(np.float32)
    ub = rng.integers(-3, V + 2, size=(rows, T)).astype(np.int32)      # empty and out-of-range slots
    ub[0, :T - 1] = ub[0, T - 1]                                      # one id several times in a history
    items = rng.integers(0, V, size=rows).astype(np.int32)
    ub[1, 0] = items[1]
    items[2] = -1
    uf = rng.random((rows, U), dtype=np.float32)
    cf = rng.random((rows, Cc), dtype=np.float32)
    y = (rng.random(rows) < 0.5).astype(np.float32)
    return om, m, E, ub, items, uf, cf, y


@pytest.mark.parametrize("kind,att,D,T", [("youtube", 0, 16, 10), ("din", 0, 16, 10), ("din", 1, 16, 10),
                                           ("youtube", 0, 64, 50), ("din", 0, 64, 7), ("din", 0, 24, 13),
                                           ("din", 1, 5, 3)])
def test_one_step_matches_oracle(oracle, kind, att, D, T):
    from goctr_amd import capi, model as gm
    U, Cc, V, B = 52, 53, 300, 256
    rows = B - 37                                                       # a padded tail inside the batch
    om, m, E, ub, items, uf, cf, y = _setup(oracle, kind, att, U, T, D, Cc, V, rows, seed=D + T)
    lr = 0.5
    loss, dE = om.emb_loss_grad(E.astype(np.float64), ub, items, uf, cf, y, B=B)
    tab = gm.EmbeddingTable(E)
    ds = gm.Dataset.ids(ub, items, uf, cf, y)
    cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0)
    m.set_embedding_training(lr)
    costs = gm.train_steps(m, ds, cfg, 1, emb=tab, want_costs=True)
    capi.sync()
    got = tab.get_rows()
    want = E.astype(np.float64) - lr * dE
    upd = np.abs(lr * dE).max()
    assert upd > 1e-5
    assert np.abs(got - want).max() <= 1e-4 * upd + 1e-7
    assert abs(float(costs[0]) - loss) < 1e-5 * max(1.0, abs(loss))
    touched = np.zeros(V, bool)
    touched[ub[(ub >= 0) & (ub < V)]] = True
    touched[items[items >= 0]] = True
    assert np.array_equal(got[~touched], E[~touched])                   # untouched rows keep their bits


SINGLES_SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
from goctr_amd import capi, model as gm
rng = np.random.default_rng(4)
U, T, D, Cc, V, B, rows = 8, 12, 16, 6, 40000, 256, 700
E = (rng.standard_normal((V, D)) * 0.5).astype(np.float32)
ub = ((rng.zipf(1.3, size=(rows, T)) - 1) %% V).astype(np.int32)        # a hot head and a long tail of singles
ub[rng.random((rows, T)) < 0.2] = -1
items = rng.integers(0, V, size=rows).astype(np.int32)
uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
y = (rng.random(rows) < 0.5).astype(np.float32)
m = gm.DinNet(U, T, D, D, Cc)
r = np.random.default_rng(1)
for n in ("mlp0", "mlp1", "mlp2"):
    m.set_weights(n, (r.standard_normal(m._shape(n)) * 0.2).astype(np.float32))
m.set_embedding_training(0.5)
tab = gm.EmbeddingTable(E); ds = gm.Dataset.ids(ub, items, uf, cf, y)
c = gm.train_steps(m, ds, capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0), 5, emb=tab, want_costs=True)   # 3 batches, wraps around
capi.sync()
np.save(%(out)r, np.concatenate([tab.get_rows().ravel(), m.get_weights("mlp0").ravel(), c]))
'''


def test_single_occurrence_ids_in_place_equals_accumulate_path(tmp_path):
    """vocabulary larger than the batch's id count: ids with one occurrence are updated in place by emb_grad
    (GOCTR_EMB_SINGLES default 1 there), all others through the fixed-point accumulators; forcing everything through
    the accumulators must give the same bits"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for singles in ("1", "0"):
        out = str(tmp_path / f"s{singles}.npy")
        env = dict(os.environ, GOCTR_EMB_SINGLES=singles, GOCTR_EMB_PLAN="0")      # (the atomics path: what widths outside the plan path use)
        r = subprocess.run([sys.executable, "-c", SINGLES_SCRIPT % dict(root=root, out=out)], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(out))
    assert np.isfinite(res[0]).all()
    assert np.array_equal(res[0], res[1])


def test_plan_path_equals_atomics_path_and_is_reproducible(tmp_path):
    """the id-major plan path (round 3: no contended atomics) against emb_grad_kernel's LDS-cache + atomics path on the same five
    steps (a hot head, a long tail of singles, wrap-around over three batches): same math, 2^-44 integer sums on both sides,
    the per-pair expression associated differently => float32 rounding apart; and the plan path twice => the same bits"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for tag, plan in (("p1", "1"), ("p2", "1"), ("a", "0")):
        out = str(tmp_path / f"{tag}.npy")
        r = subprocess.run([sys.executable, "-c", SINGLES_SCRIPT % dict(root=root, out=out)], env=dict(os.environ, GOCTR_EMB_PLAN=plan),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(out))
    assert np.isfinite(res[0]).all()
    assert np.array_equal(res[0], res[1])                               # bit-reproducible (new process, new plan build)
    assert not np.array_equal(res[0][:40000 * 16], np.zeros(40000 * 16, np.float32))
    assert np.max(np.abs(res[0] - res[2])) <= 2e-6 * max(1.0, float(np.max(np.abs(res[2]))))


@pytest.mark.parametrize("kind,att,D", [("youtube", 0, 16), ("din", 0, 16), ("din", 1, 16), ("youtube", 0, 64), ("din", 0, 64)])
def test_hot_ids_whose_runs_cross_segments_and_workgroups(oracle, kind, att, D):
    """one id carries most of the batch's pairs (a run of > 10 000 pairs: it crosses hundreds of 32-pair segments and several
    2048-pair workgroups of emb_slot_kernel -- LDS merge, atomics on the workgroup borders, emb_span_apply), a second one a
    few hundred, the rest are singles"""
    from goctr_amd import capi, model as gm
    U, Cc, V, B, T = 8, 6, 5000, 1024, 20
    rows = B
    om, m, E, ub, items, uf, cf, y = _setup(oracle, kind, att, U, T, D, Cc, V, rows, seed=5 + D)
    rng = np.random.default_rng(9)
    hot = rng.random((rows, T))
    ub[hot < 0.6] = 7
    ub[(hot >= 0.6) & (hot < 0.63)] = 4999
    items[::3] = 7
    lr = 0.5
    loss, dE = om.emb_loss_grad(E.astype(np.float64), ub, items, uf, cf, y, B=B)
    tab = gm.EmbeddingTable(E)
    ds = gm.Dataset.ids(ub, items, uf, cf, y)
    m.set_embedding_training(lr)
    costs = gm.train_steps(m, ds, capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0), 1, emb=tab, want_costs=True)
    capi.sync()
    got = tab.get_rows()
    upd = np.abs(lr * dE).max()
    assert np.abs(got - (E.astype(np.float64) - lr * dE)).max() <= 1e-4 * upd + 1e-7
    assert abs(float(costs[0]) - loss) < 1e-5 * max(1.0, abs(loss))
    touched = np.zeros(V, bool)
    touched[ub[(ub >= 0) & (ub < V)]] = True
    touched[items[items >= 0]] = True
    assert np.array_equal(got[~touched], E[~touched])
    # second step on the same batch: the accumulators the borders used were left clean
    E1 = got.copy()
    loss2, dE2 = om.emb_loss_grad(E1.astype(np.float64), ub, items, uf, cf, y, B=B)
    # (the dense weights moved too: only check finiteness and untouched rows here)
    gm.train_steps(m, ds, capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0), 1, first_batch=0, emb=tab)
    capi.sync()
    got2 = tab.get_rows()
    assert np.isfinite(got2).all() and np.array_equal(got2[~touched], E[~touched])
    assert np.abs(got2 - E1).max() <= 4 * upd                           # a sane second update, not a doubled accumulator
    del loss2, dE2


@pytest.mark.parametrize("kind", ["youtube", "din"])
def test_one_step_matches_oracle_at_cfg4_batch_size(oracle, kind):
    """batch 16 384, T = 50 -- the launch geometry of BASELINE configs[3]'s slice: the persistent-row variant of the dpv GEMM
    (gemm_nn_rows_kernel: M / 32 >= 2 x CUs), a plan of ~8 x 10^5 pairs, the four-components-per-lane slot kernel (mean
    pooling, D = 64) / emb_coef's per-pair rows (DIN, D = 16), Zipf ids"""
    from goctr_amd import capi, model as gm
    U, Cc, V, B, T = 52, 53, 50_000, 16384, 50
    D = 64 if kind == "youtube" else 16
    om, m, E, ub, items, uf, cf, y = _setup(oracle, kind, 0, U, T, D, Cc, V, B, seed=91)
    rng = np.random.default_rng(92)
    ub[:] = ((rng.zipf(1.05, size=ub.shape) - 1) % V).astype(np.int32)
    ub[rng.random(ub.shape) < 0.2] = -1
    lr = 0.5
    loss, dE = om.emb_loss_grad(E.astype(np.float64), ub, items, uf, cf, y, B=B)
    tab = gm.EmbeddingTable(E)
    ds = gm.Dataset.ids(ub, items, uf, cf, y)
    m.set_embedding_training(lr)
    costs = gm.train_steps(m, ds, capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0), 1, emb=tab, want_costs=True)
    capi.sync()
    got = tab.get_rows()
    upd = np.abs(lr * dE).max()
    assert upd > 1e-6
    assert np.abs(got - (E.astype(np.float64) - lr * dE)).max() <= 1e-4 * upd + 1e-7
    assert abs(float(costs[0]) - loss) < 1e-5 * max(1.0, abs(loss))
    touched = np.zeros(V, bool)
    touched[ub[(ub >= 0) & (ub < V)]] = True
    touched[items[items >= 0]] = True
    assert np.array_equal(got[~touched], E[~touched])


def test_one_step_matches_oracle_large_vocabulary(oracle):
    """same check as test_one_step_matches_oracle with V >> B (T+1): the in-place path for single ids is on"""
    from goctr_amd import capi, model as gm
    U, Cc, V, B, T, D = 52, 53, 60000, 256, 10, 16
    rows = B - 11
    om, m, E, ub, items, uf, cf, y = _setup(oracle, "din", 0, U, T, D, Cc, V, rows, seed=77)
    ub[5:40] = (ub[5:40] % 50)                                          # plus a block of heavily repeated ids
    lr = 0.5
    loss, dE = om.emb_loss_grad(E.astype(np.float64), ub, items, uf, cf, y, B=B)
    tab = gm.EmbeddingTable(E)
    ds = gm.Dataset.ids(ub, items, uf, cf, y)
    m.set_embedding_training(lr)
    gm.train_steps(m, ds, capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0), 1, emb=tab)
    capi.sync()
    got = tab.get_rows()
    upd = np.abs(lr * dE).max()
    assert np.abs(got - (E.astype(np.float64) - lr * dE)).max() <= 1e-4 * upd + 1e-7
    touched = np.zeros(V, bool)
    touched[ub[(ub >= 0) & (ub < V)]] = True
    touched[items[items >= 0]] = True
    assert np.array_equal(got[~touched], E[~touched])
    # a second step on the same batch works from clean marks (nothing stale from the in-place path)
    gm.train_steps(m, ds, capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0), 1, emb=tab)
    capi.sync()
    assert np.isfinite(tab.get_rows()).all() and np.array_equal(tab.get_rows()[~touched], E[~touched])


def test_reproducible_and_off_by_default(oracle):
    from goctr_amd import capi, model as gm
    U, T, D, Cc, V, B = 52, 20, 16, 53, 200, 512
    om, m, E, ub, items, uf, cf, y = _setup(oracle, "din", 0, U, T, D, Cc, V, 3 * B, seed=1)
    ds = gm.Dataset.ids(ub, items, uf, cf, y)
    cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0)
    res = []
    for lr in (0.0, 0.1, 0.1):
        m2 = gm.DinNet(U, T, D, D, Cc)
        for n in ("mlp0", "mlp1", "mlp2", "att0"):
            m2.set_weights(n, m.get_weights(n))
        tab = gm.EmbeddingTable(E)
        if lr:
            m2.set_embedding_training(lr)
        c = gm.train_steps(m2, ds, cfg, 6, emb=tab, want_costs=True)     # two passes over three batches
        capi.sync()
        res.append((tab.get_rows(), m2.get_weights("mlp0"), c))
    assert np.array_equal(res[0][0], E)                                 # frozen unless switched on
    assert not np.array_equal(res[1][0], E)
    assert np.array_equal(res[1][0], res[2][0]) and np.array_equal(res[1][1], res[2][1])   # bit-reproducible
    assert np.array_equal(res[1][2], res[2][2])


def test_training_embeddings_lowers_the_loss(oracle):
    """labels that depend on the candidate item only through its id: frozen random embeddings cannot express them
    well, trained ones can"""
    from goctr_amd import capi, model as gm
    rng = np.random.default_rng(0)
    U, T, D, Cc, V, B = 4, 5, 16, 4, 64, 1024
    rows = 8 * B
    items = rng.integers(0, V, size=rows).astype(np.int32)
    ub = rng.integers(0, V, size=(rows, T)).astype(np.int32)
    y = (items % 2).astype(np.float32)
    uf = rng.random((rows, U), dtype=np.float32)
    cf = rng.random((rows, Cc), dtype=np.float32)
    E = (rng.standard_normal((V, D)) * 0.05).astype(np.float32)
    ds = gm.Dataset.ids(ub, items, uf, cf, y)
    cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0)
    final = {}
    for lr in (0.0, 20.0):
        m = gm.YoutubeDnn(U, T, D, D, Cc)
        r = np.random.default_rng(3)
        for n in ("mlp0", "mlp1", "mlp2"):
            m.set_weights(n, (r.standard_normal(m._shape(n)) * 0.1).astype(np.float32))
        tab = gm.EmbeddingTable(E)
        m.set_embedding_training(lr)
        c = gm.train_steps(m, ds, cfg, 160, emb=tab, want_costs=True)
        capi.sync()
        assert np.all(np.isfinite(c))
        final[lr] = float(np.mean(c[-8:]))
    assert final[20.0] < 0.6 * final[0.0]


def test_errors():
    from goctr_amd import capi, model as gm
    m = gm.DinNet(4, 3, 128, 128, 4)
    with pytest.raises(capi.GoctrError):
        m.set_embedding_training(0.1)                                    # D > 64
    m = gm.DinNet(4, 3, 8, 8, 4).init_gaussian(np.random.default_rng(0))
    m.set_embedding_training(0.1)
    X = np.random.default_rng(0).random((32, 4 + 3 * 8 + 8 + 4), dtype=np.float32)
    from goctr_amd.recommend import SampleInfo
    ds = gm.Dataset.dense(X, np.zeros(32, np.float32), SampleInfo.from_dims(4, 3, 8, 4))
    with pytest.raises(capi.GoctrError):
        gm.train_steps(m, ds, capi.default_train_cfg(batch=32, epochs=1, dropout_mode=0), 1)        # dense rows carry no ids


PLAN_SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
from goctr_amd import capi, model as gm
W = %(W)d
if W > 1: capi.init_devices([0] * W)
rng = np.random.default_rng(14)
U, T, D, Cc, V, B, rows = 8, 50, 64, 6, %(V)d, %(B)d, %(rows)d
p = 1.0 / np.arange(1, V + 1) ** 1.05; p /= p.sum()
ub = rng.choice(V, size=(rows, T), p=p).astype(np.int32)
ub[rng.random((rows, T)) < 0.2] = -1
ub[3, 5] = V + 7                                                      # an id outside the table: no pair
items = rng.choice(V, size=rows, p=p).astype(np.int32)
E = (rng.standard_normal((V, D)) * 0.1).astype(np.float32)
uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
y = (rng.random(rows) < 0.5).astype(np.float32)
m = gm.YoutubeDnn(U, T, D, D, Cc)
r = np.random.default_rng(1)
for n in ("mlp0", "mlp1", "mlp2"):
    m.set_weights(n, (r.standard_normal(m._shape(n)) * 0.05).astype(np.float32))
m.set_embedding_training(0.05)
tab = gm.EmbeddingTable(E); ds = gm.Dataset.ids(ub, items, uf, cf, y)
gm.train_steps(m, ds, capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0, devices=W), 1, emb=tab)
capi.sync()
plans = [m.replica(k).emb_plan() for k in range(W)]
np.savez(%(out)r, ub=ub, items=items, **{f"{k}_{r}": v for r, pl in enumerate(plans) for k, v in pl.items()})
'''


def numpy_plan(ub, items, V, B, W, rank):
    """the plan by definition: per batch of this rank's shard the pairs with a real row, STABLE-sorted by owner-major row index"""
    rows, T = ub.shape
    Bl = B // W
    nb = -(-rows // B)
    Vw = -(-(-(-V // W)) // 4) * 4
    out = {k: [] for k in ("pair", "pslot", "pid", "slot_id", "slot_off")}
    pair_off, slot_base = [0], [0]
    for k in range(nb):
        g = k * B + rank * Bl + np.arange(Bl)                          # this rank's rows of global batch k (pad rows: no pairs)
        ids = np.full((Bl, T + 1), -1, np.int64)
        ok = g < rows
        ids[ok, :T] = ub[g[ok]]; ids[ok, T] = items[g[ok]]
        code = (np.arange(Bl)[:, None] << 12) | np.arange(T + 1)[None, :]
        ids, code = ids.ravel(), code.ravel()
        real = (ids >= 0) & (ids < V)
        ids, code = ids[real], code[real]
        pidx = ids if W == 1 else (ids % W) * Vw + ids // W
        order = np.argsort(pidx, kind="stable")
        ids, code, pidx = ids[order], code[order], pidx[order]
        head = np.ones(len(ids), bool); head[1:] = pidx[1:] != pidx[:-1]
        out["pair"].append(code); out["pid"].append(ids); out["pslot"].append(np.cumsum(head) - 1)
        out["slot_id"].append(ids[head]); out["slot_off"].append(np.concatenate([np.nonzero(head)[0], [len(ids)]]))
        pair_off.append(pair_off[-1] + len(ids)); slot_base.append(slot_base[-1] + int(head.sum()))
    res = {k: np.concatenate(v) for k, v in out.items()}
    res["pair_off"], res["slot_base"] = np.array(pair_off), np.array(slot_base)
    return res


@pytest.mark.parametrize("W,V,B,rows", [(1, 3001, 512, 1500), (1, 1000003, 2048, 4096), (4, 50021, 1024, 2500)])
def test_plan_build_is_the_stable_sort_by_row(tmp_path, W, V, B, rows):
    """the plan (csrc/emb_plan.hip: radix sort + head flags + prefix sum + fill, no atomics) is EXACTLY the stable sort of each
    batch's (sample, slot) pairs by owner-major row index -- byte for byte, a short last batch, pad slots, an id outside the
    table, W = 1 and the per-rank plans of a W = 4 run (owner-major index space, each rank its own shard); two builds in two
    processes: identical by construction (both equal the definition)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for trial in range(2):
        out = str(tmp_path / f"plan{trial}.npz")
        r = subprocess.run([sys.executable, "-c", PLAN_SCRIPT % dict(root=root, out=out, W=W, V=V, B=B, rows=rows)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        z = np.load(out)
        for rank in range(W):
            ref = numpy_plan(z["ub"], z["items"], V, B, W, rank)
            for k, v in ref.items():
                got = z[f"{k}_{rank}"]
                assert got.shape == v.shape and np.array_equal(got.astype(np.int64), v.astype(np.int64)), (trial, rank, k)
