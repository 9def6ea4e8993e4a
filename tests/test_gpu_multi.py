"""W > 1 on ONE GPU: goctr_init_devices(W, [0] * W) makes W logical ranks (engines with their own streams, arenas, model
replicas and row shards) joined by the loop-back communicator (csrc/comm.hip: RCCL rejects two ranks on one GPU), so every
piece of the multi-GPU device code -- the split step graphs around the dense all-reduce, adam_attn_kernel behind a real
reduction, emb_pack_send, the owner's base = r * Vw scan, emb_apply_gathered, the padded p * S exchange layout, item2vec's
delta exchange -- runs with W = 2, 4, 8 and is compared with the single-device run on the GLOBAL batch.

The training calls are the single-call entries: ONE goctr_train_steps / goctr_train_dataset / goctr_train_dense with
cfg.devices = W (what a single Go process' recommend.Train reaches, recommend/rcmd.go:196-246); item2vec has both the single
call (goctr_w2v_cfg.devices) and per-rank calls from W host threads (goctr_engine_select + goctr_comm_group_enable), the
sklearn-port MLP the per-rank calls only."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = r'''
import sys, ctypes as C, threading, numpy as np
sys.path.insert(0, %(root)r)
from goctr_amd import capi, model as gm
W = %(W)d
capi.init_devices([0] * W)
assert capi.engine_count() == W
'''


def run_script(body, tmp_path, tag, timeout=900, env=None, **kw):
    out = str(tmp_path / f"{tag}.npz")
    kw = dict(kw, root=ROOT, out=out)
    script = (PRELUDE + body) % kw
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", script], env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:] + "\n" + r.stderr[-4000:])
    return np.load(out)


DENSE_DP = r'''
rng = np.random.default_rng(5)
youtube = %(youtube)d
rows, U, T, D, Cc, V = 4000, 52, 50, (64 if youtube else 16), 53, 500      # 4000 rows: the 4th global batch is short (928 rows)
emb = (rng.standard_normal((V, D)) * 0.25).astype(np.float32)
ub = rng.integers(-1, V, size=(rows, T)).astype(np.int32)
it = rng.integers(0, V, size=rows).astype(np.int32)
uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
y = (rng.random(rows) < 0.5).astype(np.float32)
tab = gm.EmbeddingTable(emb); ds = gm.Dataset.ids(ub, it, uf, cf, y)
cls = gm.YoutubeDnn if youtube else gm.DinNet
names = ("mlp0", "mlp1", "mlp2") if youtube else ("mlp0", "mlp1", "mlp2", "att0")
def flat(m): return np.concatenate([m.get_weights(n).ravel() for n in names])
def mk():                      # 0.2 N(0,1): the reference's N(0,1) init saturates the output sigmoid to exactly 1.0f on some rows once dropout
    m = cls(U, T, D, D, Cc)    # rescales the activations, and BCE's 0 * log(0) is NaN there (in the reference too)
    r = np.random.default_rng(1)
    for n in ("mlp0", "mlp1", "mlp2"): m.set_weights(n, (r.standard_normal(m._shape(n)) * 0.2).astype(np.float32))
    return m
mA, mB = mk(), mk()
B = 1024
cA = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2, p0=0.01, p1=0.01, seed=7, devices=W)
cB = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2, p0=0.01, p1=0.01, seed=7, devices=1)
costA = gm.train_steps(mA, ds, cA, 13, emb=tab, want_costs=True)           # odd count: both graph parities
costA2 = gm.train_steps(mA, ds, cA, 7, first_batch=13 %% 4, emb=tab, want_costs=True)   # a second call continuing the first (cached replicas / shards)
costB = gm.train_steps(mB, ds, cB, 13, emb=tab, want_costs=True)
costB2 = gm.train_steps(mB, ds, cB, 7, first_batch=13 %% 4, emb=tab, want_costs=True)
capi.sync()
reps = np.stack([flat(mA.replica(k)) for k in range(W)])
np.savez(%(out)r, reps=reps, single=flat(mB), costA=np.concatenate([costA, costA2]), costB=np.concatenate([costB, costB2]))
'''


@pytest.mark.parametrize("W", [2, 4, 8])
@pytest.mark.parametrize("youtube", [0, 1])
def test_dense_data_parallel_one_call(tmp_path, W, youtube):
    """20 steps of the frozen-embedding step (hash dropout on), global batch 1024 over W logical ranks in ONE call: every
    replica bit-identical; equal to the single-device run on the global batch up to the summation order of the gradient
    (W partial sums instead of one: <= 1e-6 relative per step)"""
    r = run_script(DENSE_DP, tmp_path, f"dense_{W}_{youtube}", W=W, youtube=youtube)
    reps, single = r["reps"], r["single"]
    assert np.isfinite(single).all() and np.abs(single).max() > 0
    for k in range(1, W):
        assert np.array_equal(reps[0], reps[k]), f"replica {k} differs from rank 0"
    # costs: the mean BCE over the global batch
    assert np.max(np.abs(r["costA"] - r["costB"])) <= 1e-5
    # weights after 20 Adam steps (|w| ~ 1; the first step's update is lr * sign-like, so order-of-summation noise is amplified
    # only where a gradient is below Adam's eps)
    assert np.max(np.abs(reps[0] - single)) <= 2e-5 * max(1.0, float(np.abs(single).max()))
    assert np.mean(np.abs(reps[0] - single)) <= 1e-6


TRAIN_DATASET = r'''
from goctr_amd.recommend import SampleInfo
rng = np.random.default_rng(11)
U, T, D, Cc = 5, 3, 7, 5                                # model_test.go:22-27 dims
rows, B = 118 * 4, 40 * W                               # a short last batch (model_test.go:33-34: 118 @ 20)
X = rng.random((rows, U + T * D + D + Cc), dtype=np.float32)
Y = (rng.random(rows) < 0.5).astype(np.float32)
si = SampleInfo.from_dims(U, T, D, Cc)
def mk():
    m = gm.DinNet(U, T, D, D, Cc)
    r = np.random.default_rng(2)
    for n in ("mlp0", "mlp1", "mlp2"): m.set_weights(n, (r.standard_normal(m._shape(n)) * 0.3).astype(np.float32))
    return m
mA, mB = mk(), mk()
costsA = gm.Train(U, T, D, D, Cc, rows, B, 6, 3, si, X, Y, mA, dropout_seed=None, devices=W)
costsB = gm.Train(U, T, D, D, Cc, rows, B, 6, 3, si, X, Y, mB, dropout_seed=None)
names = ("mlp0", "mlp1", "mlp2", "att0")
def flat(m): return np.concatenate([m.get_weights(n).ravel() for n in names])
reps = np.stack([flat(mA.replica(k)) for k in range(W)])
np.savez(%(out)r, reps=reps, single=flat(mB), costA=costsA, costB=costsB)
'''


@pytest.mark.parametrize("W", [2, 4])
def test_model_train_dense_rows_one_call(tmp_path, W):
    """model.Train's own convention (dense TrainSample rows from the host, epochs, early stop on the last batch's cost) with
    devices = W: same epoch count, costs within 1e-5 of the single-device call, replicas bit-identical"""
    r = run_script(TRAIN_DATASET, tmp_path, f"train_{W}", W=W)
    assert len(r["costA"]) == len(r["costB"]) and len(r["costA"]) >= 1
    assert np.max(np.abs(r["costA"] - r["costB"])) <= 1e-5
    for k in range(1, W):
        assert np.array_equal(r["reps"][0], r["reps"][k])
    assert np.max(np.abs(r["reps"][0] - r["single"])) <= 2e-5


EMB_DP = r'''
rng = np.random.default_rng(9)
youtube, case = %(youtube)d, %(case)r
rows, U, T, D, Cc, V = 2048, 52, 20, (64 if youtube else 16), 53, 3001
p = 1.0 / np.arange(1, V + 1) ** 1.05; p /= p.sum()
perm = np.arange(V)
if case == "hot1":      # the hottest id is 1: its owner (1 %% W) is not rank 0
    perm[[0, 1]] = perm[[1, 0]]
ub = perm[rng.choice(V, size=(rows, T), p=p)].astype(np.int32)
it = perm[rng.choice(V, size=rows, p=p)].astype(np.int32)
if case == "empty":     # every id a multiple of W: the buckets of owners 1 .. W-1 are empty in every batch
    ub = (ub // W) * W; it = (it // W) * W
ub[rng.random((rows, T)) < 0.2] = -1
emb = (rng.standard_normal((V, D)) * 0.25).astype(np.float32)
uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
y = (rng.random(rows) < 0.5).astype(np.float32)
ds = gm.Dataset.ids(ub, it, uf, cf, y)
cls = gm.YoutubeDnn if youtube else gm.DinNet
names = ("mlp0", "mlp1", "mlp2") if youtube else ("mlp0", "mlp1", "mlp2", "att0")
def flat(m): return np.concatenate([m.get_weights(n).ravel() for n in names])
def mk():
    m = cls(U, T, D, D, Cc)
    r = np.random.default_rng(3)
    for n in ("mlp0", "mlp1", "mlp2"): m.set_weights(n, (r.standard_normal(m._shape(n)) * 0.05).astype(np.float32))
    return m.set_embedding_training(0.05)
B = 512
res = {}
for steps in (1, 6):
    tabA, tabB = gm.EmbeddingTable(emb), gm.EmbeddingTable(emb)
    mA, mB = mk(), mk()
    cA = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0, devices=W)
    cB = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=0, devices=1)
    gm.train_steps(mA, ds, cA, steps, emb=tabA)
    gm.train_steps(mB, ds, cB, steps, emb=tabB)
    capi.sync()
    res[f"tabs{steps}"] = np.stack([tabA.replica(k).get_rows() for k in range(W)])
    res[f"tab1_{steps}"] = tabB.get_rows()
    res[f"reps{steps}"] = np.stack([flat(mA.replica(k)) for k in range(W)])
    res[f"single{steps}"] = flat(mB)
    res[f"bytes{steps}"] = np.array([mA.replica(k).sparse_exchange_bytes() for k in range(W)])
res["emb0"] = emb
np.savez(%(out)r, **res)
'''


@pytest.mark.parametrize("W,youtube,case", [(2, 0, "hot1"), (4, 0, "hot1"), (2, 1, "hot1"), (2, 0, "empty"), (4, 1, "empty"), (8, 0, "zipf")])
def test_trainable_embeddings_fixed_size_exchange(tmp_path, W, youtube, case):
    """--train-emb: owner = id mod W, fixed-size buckets, three captured graphs around the collectives, with W > 1 on the
    device.  The sparse row gradients are 64-bit fixed-point sums of per-sample terms, so after ONE step the table equals the
    single-device table bit for bit (the per-sample terms do not depend on how the batch is sharded); after 6 steps the dense
    weights differ by summation order and the tables follow them.  Replicas: tables and weights bit-identical."""
    r = run_script(EMB_DP, tmp_path, f"emb_{W}_{youtube}_{case}", W=W, youtube=youtube, case=case)
    for steps in (1, 6):
        tabs, reps = r[f"tabs{steps}"], r[f"reps{steps}"]
        assert np.isfinite(tabs).all()
        assert np.abs(tabs[0] - r["emb0"]).max() > 0, "the table did not move"
        for k in range(1, W):
            assert np.array_equal(tabs[0], tabs[k]), f"table replica {k} differs"
            assert np.array_equal(reps[0], reps[k]), f"weight replica {k} differs"
        assert (r[f"bytes{steps}"] > 0).all()
    # one step: exact integer sums of identical per-sample terms (2^-44 is one unit of the fixed-point accumulator); what may
    # differ is the float rounding of  E - lr * sum  (applied in place on one device, as a gathered float delta on W)
    d1 = np.abs(r["tabs1"][0].astype(np.float64) - r["tab1_1"].astype(np.float64)).max()
    assert d1 <= 6e-8, f"after one step the table differs from the single-device table by {d1}"
    d6 = np.abs(r["tabs6"][0] - r["tab1_6"]).max()
    assert d6 <= 2e-6, d6
    assert np.abs(r["reps6"][0] - r["single6"]).max() <= 2e-5


W2V_THREADS = r'''
from goctr_amd import embedding as ge
rng = np.random.default_rng(3)
V, n, dim = 60, 4000, 16
p = 1.0 / np.arange(1, V + 1); p /= p.sum()
docs = [rng.choice(V, size=n, p=p).astype(np.int32) for _ in range(W)]
counts = np.bincount(np.concatenate(docs), minlength=V) + 1
p0 = (rng.random((V, dim)) - 0.5) / dim
def one(doc):                      # a rank's pass on its own (single device): p_r
    m = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=True)
    m.create(counts, p0)
    lr = m.train_pass(doc, n * W, None, lr=0.025)
    return m.get_param(), m.get_aux(), lr
solo = [one(d) for d in docs]
out = [None] * W
errs = []
def rank(k):
    try:
        capi.engine_select(k)
        capi.comm_group_enable(True)
        m = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=True)
        m.create(counts, p0)
        lr = m.train_pass(docs[k], n * W, None, lr=0.025)
        out[k] = (m.get_param(), m.get_aux(), lr)
    except Exception as e:
        errs.append(repr(e))
ths = [threading.Thread(target=rank, args=(k,)) for k in range(W)]
[t.start() for t in ths]; [t.join() for t in ths]
assert not errs, errs
np.savez(%(out)r, solo_p=np.stack([s[0] for s in solo]), solo_a=np.stack([s[1] for s in solo]), p0=p0,
         dp_p=np.stack([o[0] for o in out]), dp_a=np.stack([o[1] for o in out]))
'''


W2V_SINGLE_CALL = r'''
from goctr_amd import embedding as ge
rng = np.random.default_rng(3)
V, n, dim = 60, 4000, 16
p = 1.0 / np.arange(1, V + 1); p /= p.sum()
docs = [rng.choice(V, size=n, p=p).astype(np.int32) for _ in range(W)]
keeps = [(rng.random(n) < 0.8).astype(np.uint8) for _ in range(W)]
counts = np.bincount(np.concatenate(docs), minlength=V) + 1
p0 = (rng.random((V, dim)) - 0.5) / dim
def threads():         # the per-rank calls of test_item2vec_delta_exchange, three passes on the same handles: the expected result
    out = [[None] * W for _ in range(3)]; errs = []; bar = threading.Barrier(W)
    def rank(k):
        try:
            capi.engine_select(k)
            capi.comm_group_enable(True)
            m = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=True)
            m.create(counts, p0)
            lr = m.train_pass(docs[k], n * W, keeps[k], lr=0.025)
            out[0][k] = (m.get_param(), m.get_aux(), lr)
            bar.wait()
            lr = m.train_resident(n * W, lr=out[0][0][2])            # (the single call hands rank 0's rate to every rank)
            out[1][k] = (m.get_param(), m.get_aux(), lr)
            m.set_param(p0); m.set_aux(np.zeros_like(out[1][k][1]))
            lr = m.train_resident(n * W, lr=0.025)                   # (the window-shrink generators carry on from pass 2)
            out[2][k] = (m.get_param(), m.get_aux(), lr)
            capi.comm_group_enable(False)
        except Exception as e:
            errs.append(repr(e)); bar.abort()
    ths = [threading.Thread(target=rank, args=(k,)) for k in range(W)]
    [t.start() for t in ths]; [t.join() for t in ths]
    assert not errs, errs
    return out
exp1, exp2, exp3 = threads()
capi.engine_select(0)
m = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=True, devices=W)
m.create(counts, p0)
lr1 = m.train_pass(np.concatenate(docs), n * W, np.concatenate(keeps), lr=0.025)       # ONE call: shards, replicas, exchange
g1 = (m.get_param(), m.get_aux())
lr2 = m.train_resident(n * W)                                                            # second pass on the resident shards
g2 = (m.get_param(), m.get_aux())
# a set_param from outside puts the replicas out of date: the next pass must broadcast again
m.set_param(p0); m.set_aux(np.zeros_like(g1[1]))
lr3 = m.train_resident(n * W, lr=0.025)
g3 = (m.get_param(), m.get_aux())
# the device-resident corpus path (TrainEmbedding over item ids): dictionary, IndexedDoc and subsampling mask are made on engine 0,
# the ranks take their shards device to device -- against the host-doc path fed with the same doc and mask
ids = [rng.choice(V, size=3000, p=p).astype(np.int64) + 1000 for _ in range(4)]
solo = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=True, min_count=1, rng=np.random.default_rng(9))
solo.TrainIds(ids, 12000, seed=5)
cdoc = solo.corpus.IndexedDoc(); ckeep = solo.keep_mask(cdoc.size)
cps_cfs = np.asarray(solo.corpus.Dictionary()[1], np.int64)
cp0 = (np.random.default_rng(9).random((solo.V, dim)) - 0.5) / dim
h = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=True, devices=W)
h.create(cps_cfs, cp0)
h.train_pass(cdoc, solo.corpus.Len(), ckeep, lr=0.025)
c = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=True, min_count=1, rng=np.random.default_rng(9), devices=W)
c.TrainIds(ids, 12000, seed=5)
np.savez(%(out)r, e1p=exp1[0][0], e1a=exp1[0][1], e2p=exp2[0][0], e2a=exp2[0][1], e3p=exp3[0][0], e3a=exp3[0][1], g1p=g1[0], g1a=g1[1], g2p=g2[0], g2a=g2[1],
         g3p=g3[0], g3a=g3[1], lrs=np.array([exp1[0][2], exp2[0][2], lr1, lr2, lr3, exp3[0][2]]), p0=p0,
         hp=h.get_param(), ha=h.get_aux(), cp=c.get_param(), ca=c.get_aux(), sp=solo.get_param())
'''


@pytest.mark.parametrize("W", [2, 4])
def test_item2vec_single_call(tmp_path, W):
    """goctr_w2v_cfg.devices = W: ONE goctr_w2v_train call cuts the doc at the slice boundaries, builds the replicas, broadcasts,
    trains every shard and sums the deltas -- bit-equal to the W per-rank calls from W threads (same kernels, same fixed-order
    reduction), over two passes, and again after the caller overwrote the vectors."""
    r = run_script(W2V_SINGLE_CALL, tmp_path, f"w2v1_{W}", W=W)
    assert np.array_equal(r["g1p"], r["e1p"]) and np.array_equal(r["g1a"], r["e1a"])
    assert np.array_equal(r["g2p"], r["e2p"]) and np.array_equal(r["g2a"], r["e2a"])
    assert np.array_equal(r["g3p"], r["e3p"]) and np.array_equal(r["g3a"], r["e3a"])
    assert r["lrs"][2] == r["lrs"][0] and r["lrs"][3] == r["lrs"][1] and r["lrs"][4] == r["lrs"][5]
    assert np.max(np.abs(r["g1p"] - r["p0"])) > 1e-4 and not np.array_equal(r["g2p"], r["g1p"])
    assert np.array_equal(r["cp"], r["hp"]) and np.array_equal(r["ca"], r["ha"])
    assert not np.array_equal(r["cp"], r["sp"])


W2V_HOGWILD_MULTI = r'''
from goctr_amd import embedding as ge
rng = np.random.default_rng(3)
V, dim = 200, 16
sessions = []
for _ in range(8000):
    g = rng.integers(0, 2)
    sessions.append(rng.integers(g * 100, g * 100 + 100, size=50))
doc = np.concatenate(sessions).astype(np.int32)
counts = np.bincount(doc, minlength=V)
capi.engine_select(0)
m = ge.Word2Vec(dim=dim, deterministic=False, streams=512, rng=np.random.default_rng(4), devices=W)
m.create(counts)
p0 = m.get_param()
for _ in range(3):
    m.train_pass(doc, doc.size, None, lr=0.025)
np.savez(%(out)r, p=m.get_param(), p0=p0)
'''


@pytest.mark.parametrize("W", [2, 4])
def test_item2vec_single_call_hogwild(tmp_path, W):
    """the Hogwild kernel under goctr_w2v_cfg.devices = W (every rank its own shard and hot-row copies, deltas summed after
    the pass): two disjoint session vocabularies must come apart as they do on one device"""
    r = run_script(W2V_HOGWILD_MULTI, tmp_path, f"w2vh_{W}", W=W)
    P = r["p"]
    assert np.all(np.isfinite(P)) and np.max(np.abs(P - r["p0"])) > 1e-3
    Pn = P / np.linalg.norm(P, axis=1, keepdims=True)
    S = Pn @ Pn.T
    within = (S[:100, :100].sum() - 100) / (100 * 99)
    across = S[:100, 100:].mean()
    assert within > across + 0.2, (within, across)


W2V_SEGMENTED_RULE = r'''
from goctr_amd import embedding as ge
sys.path.insert(0, %(root)r)
from oracle import pyoracle as o
rng = np.random.default_rng(8)
V, n, dim, nseg = 60, 4000, 8, 3
doc = np.concatenate([rng.integers(0, 30, size=n // 2), rng.integers(20, V, size=n // 2)]).astype(np.int32)   # rank 0 never sees words >= 30
counts = np.bincount(doc, minlength=V) + 1
p0 = (rng.random((V, dim)) - 0.5) / dim
capi.engine_select(0)
m = ge.Word2Vec(dim=dim, deterministic=False, streams=1, slices=1, rng=np.random.default_rng(4), devices=W)
m.create(counts, param0=p0)
m.train_pass(doc, n, None, lr=0.025)
cuts = np.zeros(W + 1, np.int64)
capi.check(capi.load().goctr_w2v_shard_cuts(C.c_int64(n), C.c_int(1), C.c_int(W), capi.ptr(cuts, C.c_int64)))
# the rule, evaluated by the oracle: every rank walks segment s of ITS shard from the common snapshot (windows clipped at the shard's
# ends, a fresh window-shrink stream per (rank, segment): csrc/w2v.hip seeds 1 + K ((rank << 20) + (segment << 32))), then every
# row moves by the sum of the ranks' deltas divided by the number of ranks that changed it
cfg = o.w2v_cfg(dim=dim, optimizer="hs")
paths = o.huffman_paths(counts)
sig = o.sigmoid_table()
P, A = p0.copy(), np.zeros((V - 1, dim))
K = 0x9E3779B97F4A7C15
for s_ in range(nseg):
    ds = []
    for r in range(W):
        lo, hi = int(cuts[r]), int(cuts[r + 1])
        a0, a1 = lo + (hi - lo) * s_ // nseg, lo + (hi - lo) * (s_ + 1) // nseg
        p, a = P.copy(), A.copy()
        lcg = o.Lcg((1 + K * ((r << 20) + (s_ << 32))) %% (1 << 64))
        o.w2v_train_range(cfg, doc, lo, hi, a0, a1, p, a, paths, sig, lcg, 0.025, 0, n)
        ds.append((p - P, a - A))
    for M, k in ((P, 0), (A, 1)):
        d = sum(x[k] for x in ds)
        cnt = sum((np.abs(x[k]).max(1) > 0).astype(np.float64) for x in ds)
        M += d / np.maximum(cnt, 1.0)[:, None]
np.savez(%(out)r, p=m.get_param(), a=m.get_aux(), ep=P, ea=A, p0=p0)
'''


def test_item2vec_segmented_exchange_matches_the_rule_per_segment(tmp_path):
    """the Hogwild passes' exchange (csrc/w2v.hip exchange_deltas(avg): snapshot, w2v_touched_kernel, the all-reduce of
    deltas and touched flags, w2v_apply_avg_kernel, the segmented launches) on TWO logical ranks with ONE stream each -- nothing to
    race with, so the device must reproduce the rule evaluated by the oracle segment by segment (three segments, GOCTR_W2V_SEGMENTS=3;
    words >= 30 only occur in rank 1's shard: those rows keep their whole update)"""
    r = run_script(W2V_SEGMENTED_RULE, tmp_path, "w2vseg", W=2, env={"GOCTR_W2V_SEGMENTS": "3"})
    assert np.max(np.abs(r["ep"] - r["p0"])) > 1e-3
    assert np.max(np.abs(r["p"] - r["ep"])) <= 1e-9 and np.max(np.abs(r["a"] - r["ea"])) <= 1e-9
    assert np.max(np.abs(r["p"][40:] - r["p0"][40:])) > 1e-4


@pytest.mark.parametrize("W", [2, 4])
def test_item2vec_delta_exchange(tmp_path, W):
    """item2vec, W ranks from W host threads: snapshot, local deterministic pass on the rank's corpus shard, all-reduce of
    the parameter deltas, p = p0 + sum_r (p_r - p0).  The local passes are bit-exact single-stream f64, so the expected
    result is computable from W single-device passes."""
    r = run_script(W2V_THREADS, tmp_path, f"w2v_{W}", W=W)
    for k in range(1, W):
        assert np.array_equal(r["dp_p"][0], r["dp_p"][k]) and np.array_equal(r["dp_a"][0], r["dp_a"][k])
    exp_p = r["p0"] + sum(r["solo_p"][k] - r["p0"] for k in range(W))
    exp_a = sum(r["solo_a"][k] for k in range(W))          # (the HS node vectors start at zero)
    assert np.max(np.abs(r["dp_p"][0] - exp_p)) <= 1e-12
    assert np.max(np.abs(r["dp_a"][0] - exp_a)) <= 1e-12
    assert np.max(np.abs(r["dp_p"][0] - r["p0"])) > 1e-4


MLP_THREADS = r"""
from goctr_amd import mlp as gmlp
rng = np.random.default_rng(4)
F, H, B, rows = 37, 20, 128, 1024
X = rng.random((rows, F), dtype=np.float32)
y = (X[:, 0] + X[:, 1] > 1.0).astype(np.float32)
units = [F, H, 1]
def make(batch):
    clf = gmlp.MLPClassifier([H], "relu", "adam", 1e-4)
    clf.BatchSize = batch
    clf.create(units, batch, clf.init_params(units, np.random.default_rng(1)))
    return clf
single = make(B); single.upload(X, y); single.train_steps(11); capi.sync()
Bl = B // W
out = [None] * W; errs = []
def rank(k):
    try:
        capi.engine_select(k)
        capi.comm_group_enable(True)
        idx = np.concatenate([np.arange(b * B + k * Bl, b * B + (k + 1) * Bl) for b in range(rows // B)])   # rank k's rows of every global batch
        clf = make(Bl); clf.upload(X[idx], y[idx]); clf.train_steps(11); capi.sync()
        out[k] = clf.get_params()
    except Exception as e:
        errs.append(repr(e))
ths = [threading.Thread(target=rank, args=(k,)) for k in range(W)]
[t.start() for t in ths]; [t.join() for t in ths]
assert not errs, errs
np.savez(%(out)r, single=single.get_params(), dp=np.stack(out))
"""


@pytest.mark.parametrize("W", [2, 4])
def test_sklearn_mlp_data_parallel(tmp_path, W):
    """the sklearn-port MLP (f64): local slab sums with the GLOBAL batch in the 1/n factors, one f64 all-reduce of
    [G | loss-term sum], the identical per-parameter Adam everywhere -- W ranks from W host threads vs the single-device
    step on the global batch (f64 summation order only)"""
    r = run_script(MLP_THREADS, tmp_path, f"mlp_{W}", W=W)
    for k in range(1, W):
        assert np.array_equal(r["dp"][0], r["dp"][k])
    assert np.isfinite(r["single"]).all()
    assert np.max(np.abs(r["dp"][0] - r["single"])) <= 1e-9 * max(1.0, float(np.abs(r["single"]).max()))


ABORT = r'''
rng = np.random.default_rng(5)
rows, U, T, D, Cc, V = 1024, 52, 10, 16, 53, 100
emb = (rng.standard_normal((V, D)) * 0.25).astype(np.float32)
ub = rng.integers(-1, V, size=(rows, T)).astype(np.int32)
it = rng.integers(0, V, size=rows).astype(np.int32)
uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
y = (rng.random(rows) < 0.5).astype(np.float32)
# per-rank calls from W threads; rank 1 never makes its call: rank 0 must come back with an error, not hang
out = {}
def rank(k):
    capi.engine_select(k)
    capi.comm_group_enable(True)
    tab = gm.EmbeddingTable(emb); ds = gm.Dataset.ids(ub, it, uf, cf, y)
    m = gm.DinNet(U, T, D, D, Cc).init_gaussian(np.random.default_rng(1))
    cfg = capi.default_train_cfg(batch=256, epochs=1, dropout_mode=0)
    if k == 1:
        out[k] = "skipped"
        # (its handles go away while rank 0 captures its step graphs: goctr_model_destroy once waited with hipDeviceSynchronize,
        # which invalidated the capture in one run of 14; the destroy paths now wait on the engine's own streams only)
        return
    try:
        gm.train_steps(m, ds, cfg, 3, emb=tab, want_costs=True)
        out[k] = "returned"
    except capi.GoctrError as e:
        out[k] = "error: " + str(e)
ths = [threading.Thread(target=rank, args=(k,)) for k in range(W)]
[t.start() for t in ths]; [t.join() for t in ths]
np.savez(%(out)r, msg=np.array([out[0]]))
'''


def test_missing_rank_fails_instead_of_hanging(tmp_path):
    """a rank that never reaches the collective: the loop-back barrier's timeout aborts the group and the waiting rank's call
    returns an error (the watchdog the advisor asked for, on the communicator these boxes can run)"""
    r = run_script(ABORT, tmp_path, "abort", W=2, env={"GOCTR_LOOP_TIMEOUT_S": "3"}, timeout=300)
    msg = str(r["msg"][0])
    assert msg.startswith("error:") and "did not reach the collective" in msg, msg


RCCL_GROUP_ONE = r'''
rng = np.random.default_rng(5)
rows, U, T, D, Cc, V = 4000, 52, 50, 16, 53, 500
emb = (rng.standard_normal((V, D)) * 0.25).astype(np.float32)
ub = rng.integers(-1, V, size=(rows, T)).astype(np.int32)
it = rng.integers(0, V, size=rows).astype(np.int32)
uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
y = (rng.random(rows) < 0.5).astype(np.float32)
names = ("mlp0", "mlp1", "mlp2", "att0")
def flat(m): return np.concatenate([m.get_weights(n).ravel() for n in names])
def mk():
    m = gm.DinNet(U, T, D, D, Cc)
    r = np.random.default_rng(1)
    for n in ("mlp0", "mlp1", "mlp2"): m.set_weights(n, (r.standard_normal(m._shape(n)) * 0.2).astype(np.float32))
    return m
res = {}
for emb_lr in (0.0, 0.05):
    tabA, tabB = gm.EmbeddingTable(emb), gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, y)
    mA, mB = mk(), mk()
    if emb_lr: mA.set_embedding_training(emb_lr); mB.set_embedding_training(emb_lr)
    cA = capi.default_train_cfg(batch=1024, epochs=1, dropout_mode=2, p0=0.01, p1=0.01, seed=7, devices=1)    # the multi-device entry, one rank
    cB = capi.default_train_cfg(batch=1024, epochs=1, dropout_mode=2, p0=0.01, p1=0.01, seed=7, devices=0)    # the plain call
    cA_ = gm.train_steps(mA, ds, cA, 9, emb=tabA, want_costs=True)
    cB_ = gm.train_steps(mB, ds, cB, 9, emb=tabB, want_costs=True)
    capi.sync()
    res[f"wA{emb_lr}"], res[f"wB{emb_lr}"] = flat(mA), flat(mB)
    res[f"tA{emb_lr}"], res[f"tB{emb_lr}"] = tabA.get_rows(), tabB.get_rows()
    res[f"cA{emb_lr}"], res[f"cB{emb_lr}"] = cA_, cB_
mode = C.c_int(0)
capi.check(capi.load().goctr_comm_capture_mode(C.byref(mode)))
res["mode"] = np.array([mode.value])
np.savez(%(out)r, **res)
'''


def test_single_call_entry_over_an_rccl_group_of_one(tmp_path):
    """mode 2 (goctr_init_devices over distinct devices: ncclCommInitAll, every rank a thread) as far as a one-GPU box can run
    it: a group of ONE engine with GOCTR_FORCE_COMM=1 -- ncclCommInitAll, ncclBroadcast of the replica state, the shard scatter as
    grouped ncclSend / ncclRecv, the split step graphs around ncclAllReduce (captured after the self-test), the fixed-size sparse
    exchange -- must reproduce the plain single-device call bit for bit (one rank: every collective is the identity)"""
    r = run_script(RCCL_GROUP_ONE, tmp_path, "rccl1", W=1, env={"GOCTR_FORCE_COMM": "1"})
    for lr in ("0.0", "0.05"):
        assert np.isfinite(r[f"wB{lr}"]).all()
        assert np.array_equal(r[f"cA{lr}"], r[f"cB{lr}"])
        assert np.array_equal(r[f"wA{lr}"], r[f"wB{lr}"])
        assert np.array_equal(r[f"tA{lr}"], r[f"tB{lr}"])
    assert np.abs(r["tA0.05"] - r["tA0.0"]).max() > 0


HOUSEKEEPING = r'''
rng = np.random.default_rng(8)
U, T, D, Cc, V = 52, 10, 16, 53, 200
def data(rows, seed):
    r = np.random.default_rng(seed)
    return (r.integers(-1, V, size=(rows, T)).astype(np.int32), r.integers(0, V, size=rows).astype(np.int32),
            r.random((rows, U), dtype=np.float32), r.random((rows, Cc), dtype=np.float32), (r.random(rows) < 0.5).astype(np.float32))
emb = (rng.standard_normal((V, D)) * 0.25).astype(np.float32)
names = ("mlp0", "mlp1", "mlp2", "att0")
def flat(m): return np.concatenate([m.get_weights(n).ravel() for n in names])
def mk(seed=1):
    m = gm.DinNet(U, T, D, D, Cc)
    r = np.random.default_rng(seed)
    for n in ("mlp0", "mlp1", "mlp2"): m.set_weights(n, (r.standard_normal(m._shape(n)) * 0.2).astype(np.float32))
    return m
tab = gm.EmbeddingTable(emb)
ds1, ds2 = gm.Dataset.ids(*data(1000, 1)), gm.Dataset.ids(*data(777, 2))
cW = capi.default_train_cfg(batch=64 * W, epochs=1, dropout_mode=0, devices=W)
c1 = capi.default_train_cfg(batch=64 * W, epochs=1, dropout_mode=0, devices=1)
log = {}
# 1. a call that must fail on every rank before any collective (batch not a multiple of devices), then a good one: the group is usable again
bad = capi.default_train_cfg(batch=64 * W + 1, epochs=1, dropout_mode=0, devices=W)
mA, mB = mk(), mk()
try:
    gm.train_steps(mA, ds1, bad, 2, emb=tab); log["bad_batch"] = "no error"
except capi.GoctrError as e:
    log["bad_batch"] = str(e)
gm.train_steps(mA, ds1, cW, 3, emb=tab); gm.train_steps(mB, ds1, c1, 3, emb=tab)
# 2. weights changed between two multi-device calls: the replicas must be re-broadcast
w = mA.get_weights("mlp1") * 0.5
mA.set_weights("mlp1", w); mB.set_weights("mlp1", w)
gm.train_steps(mA, ds1, cW, 3, first_batch=3, emb=tab); gm.train_steps(mB, ds1, c1, 3, first_batch=3, emb=tab)
# 3. another dataset (new shards), a ragged one; then back to the first (shards rebuilt)
gm.train_steps(mA, ds2, cW, 4, emb=tab); gm.train_steps(mB, ds2, c1, 4, emb=tab)
gm.train_steps(mA, ds1, cW, 2, emb=tab); gm.train_steps(mB, ds1, c1, 2, emb=tab)
# 4. a plain single-device call on the model that has replicas, then multi again (re-broadcast)
gm.train_steps(mA, ds1, c1, 2, first_batch=2, emb=tab); gm.train_steps(mB, ds1, c1, 2, first_batch=2, emb=tab)
gm.train_steps(mA, ds1, cW, 2, first_batch=4, emb=tab); gm.train_steps(mB, ds1, c1, 2, first_batch=4, emb=tab)
capi.sync()
reps = np.stack([flat(mA.replica(k)) for k in range(W)])
# 5. the table changed through set_rows: the table replicas follow on the next call (frozen embeddings: only the forward reads them)
e2 = emb.copy(); e2[:50] *= -1.0
tab.set_rows(e2) if hasattr(tab, "set_rows") else capi.check(capi.load().goctr_emb_set_rows(tab._h, C.c_int64(0), C.c_int64(V), capi.ptr(capi.f32(e2), C.c_float)))
gm.train_steps(mA, ds1, cW, 1, emb=tab)
capi.sync()
tabs = np.stack([tab.replica(k).get_rows() for k in range(W)])
np.savez(%(out)r, reps=reps, single=flat(mB), tabs=tabs, e2=e2, msg=np.array([log["bad_batch"]]))
'''


@pytest.mark.parametrize("W", [2, 4])
def test_replica_and_shard_caches_follow_what_changed(tmp_path, W):
    """the single-call entry caches replicas and shards on the handles: an argument error before any collective leaves the group
    usable; set_weights between two calls, another dataset, a plain single-device call in between, set_rows on the table -- every
    change is re-broadcast / re-sharded, the replicas stay bit-identical and track the single-device run"""
    r = run_script(HOUSEKEEPING, tmp_path, f"house_{W}", W=W)
    assert "not a multiple" in str(r["msg"][0])
    for k in range(1, W):
        assert np.array_equal(r["reps"][0], r["reps"][k])
        assert np.array_equal(r["tabs"][k], r["e2"])
    assert np.array_equal(r["tabs"][0], r["e2"])
    assert np.isfinite(r["single"]).all()
    assert np.max(np.abs(r["reps"][0] - r["single"])) <= 2e-5


def test_bench_single_process_two_logical_ranks():
    """bench.py --gpus 2 --single-process: ONE process, goctr_init_devices + cfg.devices = 2 (here two logical ranks on device 0:
    the loop-back communicator) -- the line the driver's SCALE run can take for mode 2, replicas checked bit for bit"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["GOCTR_BENCH_DEVICES"] = "0,0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process", "--steps", "6", "--warmup", "2",
                        "--rows", "32768", "--regions", "3"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 * 8192 and d["config"]["parallelism"] == "dp2"
    assert d["replicas_bit_identical"] is True and d["value"] > 0 and len(d["timed_regions_ms"]) == 3
    # (round 6) every rank's own device time of the median region, like the per-process wall times of the one-process-per-GPU line
    per = d["per_rank_ms_per_step"]
    assert len(per) == 2 and all(0 < p <= d["ms_per_step"] * 1.05 for p in per), (per, d["ms_per_step"])
