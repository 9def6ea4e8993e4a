"""The multi-GPU step path (reduce -> RCCL all-reduce -> Adam, two captured graphs per state parity) exercised on ONE
GPU: GOCTR_FORCE_COMM=1 makes goctr_comm_init build a real one-rank RCCL communicator, so every call the 8-GPU run
makes (dlopen, ncclGetUniqueId, ncclCommInitRank, ncclAllReduce on the engine stream between the two graphs) runs
here; with one rank the all-reduce is the identity, so the result must equal the fused single-GPU path bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, ctypes as C, numpy as np
sys.path.insert(0, %(root)r)
from goctr_amd import capi, model as gm
capi.init(0)
L = capi.load()
if %(comm)d:
    idbuf = (C.c_uint8 * 128)()
    capi.check(L.goctr_comm_unique_id(idbuf))
    capi.check(L.goctr_comm_init(C.c_int(0), C.c_int(1), idbuf))
rng = np.random.default_rng(5)
rows, U, T, D, Cc, V = 4096, 52, 50, %(D)d, 53, 500
emb = (rng.standard_normal((V, D)) * 0.25).astype(np.float32)
ub = rng.integers(-1, V, size=(rows, T)).astype(np.int32)
it = rng.integers(0, V, size=rows).astype(np.int32)
uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
y = (rng.random(rows) < 0.5).astype(np.float32)
tab = gm.EmbeddingTable(emb); ds = gm.Dataset.ids(ub, it, uf, cf, y)
m = (gm.YoutubeDnn if %(youtube)d else gm.DinNet)(U, T, D, D, Cc).init_gaussian(np.random.default_rng(1))
cfg = capi.default_train_cfg(batch=1024, epochs=1, dropout_mode=0)
gm.train_steps(m, ds, cfg, 7, emb=tab)          # odd count: both graph parities are replayed
gm.train_steps(m, ds, cfg, 4, first_batch=3, emb=tab)   # a second call that continues where the first ended (7 %% 4 batches)
capi.sync()
out = np.concatenate([m.get_weights(n).ravel() for n in (("mlp0", "mlp1", "mlp2") if %(youtube)d else ("mlp0", "mlp1", "mlp2", "att0"))])
np.save(%(out)r, out)
if %(comm)d:
    mode = C.c_int(0)
    capi.check(L.goctr_comm_capture_mode(C.byref(mode)))
    open(%(out)r + ".mode", "w").write(str(mode.value))
    capi.check(L.goctr_comm_destroy())
'''


def run(tmp_path, comm, graph, youtube=False, capture=None):
    out = str(tmp_path / f"w_{comm}_{graph}_{youtube}_{capture}.npy")
    env = dict(os.environ)
    env["GOCTR_FORCE_COMM"] = "1" if comm else "0"
    if capture is not None:
        env["GOCTR_DP_CAPTURE_COMM"] = str(capture)
    if not graph:
        env["GOCTR_NO_GRAPH"] = "1"
    r = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT, comm=int(comm), out=out, youtube=int(youtube),
                                                            D=64 if youtube else 16)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("graph,youtube", [(True, False), (False, False), (True, True)])
def test_one_rank_communicator_equals_fused_path(tmp_path, graph, youtube):
    """(graph: the data-parallel step is pipelined -- adam_attn_kernel, b(n) + a(n + 1) as one graph -- and the second call is a
    carried start; DIN with its att0 hand-over and YouTube-DNN, whose attention does not depend on the weights)"""
    ref = run(tmp_path, comm=False, graph=graph, youtube=youtube)
    got = run(tmp_path, comm=True, graph=graph, youtube=youtube, capture=0)        # the all-reduce between graph launches
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0
    assert np.array_equal(ref, got)
    if graph:
        # the all-reduce as a node of the 4- / 2-step graphs (after the communicator's self-test: captured vs eager collective)
        cap = run(tmp_path, comm=True, graph=True, youtube=youtube, capture=1)
        mode = int(open(str(tmp_path / f"w_True_True_{youtube}_1.npy") + ".mode").read())
        assert mode == 1, "the captured RCCL all-reduce did not pass its self-test on this box"
        assert np.array_equal(ref, cap)


W2V_SCRIPT = r'''
import sys, ctypes as C, numpy as np
sys.path.insert(0, %(root)r)
from goctr_amd import capi, embedding as ge
capi.init(0)
L = capi.load()
if %(comm)d:
    idbuf = (C.c_uint8 * 128)()
    capi.check(L.goctr_comm_unique_id(idbuf))
    capi.check(L.goctr_comm_init(C.c_int(0), C.c_int(1), idbuf))
rng = np.random.default_rng(3)
V, n, dim = 60, 4000, 16
p = 1.0 / np.arange(1, V + 1); p /= p.sum()
doc = rng.choice(V, size=n, p=p).astype(np.int32)
counts = np.bincount(doc, minlength=V) + 1
p0 = (rng.random((V, dim)) - 0.5) / dim
m = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=True)
m.create(counts, p0)
lr = m.train_pass(doc, n, None, lr=0.025)
lr = m.train_pass(doc, n, None, lr=lr)
np.save(%(out)r, np.concatenate([m.get_param().ravel(), m.get_aux().ravel(), [lr]]))
'''


def test_item2vec_delta_exchange_one_rank(tmp_path):
    """item2vec's data-parallel step (snapshot, pass, all-reduce of the parameter deltas, p = p0 + sum) with a one-rank
    RCCL communicator: p0 + (p - p0) must give back the single-GPU result (deterministic mode) up to the rounding of
    that one subtraction / addition"""
    res = []
    for comm in (0, 1):
        out = str(tmp_path / f"w2v_{comm}.npy")
        env = dict(os.environ)
        env["GOCTR_FORCE_COMM"] = str(comm)
        r = subprocess.run([sys.executable, "-c", W2V_SCRIPT % dict(root=ROOT, comm=comm, out=out)], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(out))
    assert np.isfinite(res[0]).all()
    assert np.max(np.abs(res[0] - res[1])) <= 1e-12          # (the rounding of p0 + (p - p0), carried through the second pass)


MLP_SCRIPT = r'''
import sys, ctypes as C, numpy as np
sys.path.insert(0, %(root)r)
from goctr_amd import capi, mlp as gmlp
capi.init(0)
L = capi.load()
if %(comm)d:
    idbuf = (C.c_uint8 * 128)()
    capi.check(L.goctr_comm_unique_id(idbuf))
    capi.check(L.goctr_comm_init(C.c_int(0), C.c_int(1), idbuf))
rng = np.random.default_rng(4)
F, H, B, rows = 37, 20, 128, 1024
X = rng.random((rows, F), dtype=np.float32)
y = (X[:, 0] + X[:, 1] > 1.0).astype(np.float32)
clf = gmlp.MLPClassifier([H], "relu", "adam", 1e-4)
clf.BatchSize = B
units = [F, H, 1]
clf.create(units, B, clf.init_params(units, np.random.default_rng(1)))
clf.upload(X, y)
clf.train_steps(11)
capi.sync()
np.save(%(out)r, clf.get_params())
'''


def test_mlp_data_parallel_step_one_rank(tmp_path):
    """the sklearn-port MLP's split step (slab sums -> f64 all-reduce of [G | loss-term sum] -> update) with a one-rank RCCL
    communicator must reproduce the fused single-GPU step bit for bit"""
    res = []
    for comm in (0, 1):
        out = str(tmp_path / f"mlp_{comm}.npy")
        env = dict(os.environ)
        env["GOCTR_FORCE_COMM"] = str(comm)
        r = subprocess.run([sys.executable, "-c", MLP_SCRIPT % dict(root=ROOT, comm=comm, out=out)], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(out))
    assert np.isfinite(res[0]).all() and np.array_equal(res[0], res[1])


EMB_SCRIPT = r'''
import sys, ctypes as C, numpy as np
sys.path.insert(0, %(root)r)
from goctr_amd import capi, model as gm
capi.init(0)
L = capi.load()
if %(comm)d:
    idbuf = (C.c_uint8 * 128)()
    capi.check(L.goctr_comm_unique_id(idbuf))
    capi.check(L.goctr_comm_init(C.c_int(0), C.c_int(1), idbuf))
rng = np.random.default_rng(5)
rows, U, T, D, Cc, V = 2048, 52, 20, 16, 53, 300
emb = (rng.standard_normal((V, D)) * 0.25).astype(np.float32)
ub = rng.integers(-1, V, size=(rows, T)).astype(np.int32)
it = rng.integers(0, V, size=rows).astype(np.int32)
uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
y = (rng.random(rows) < 0.5).astype(np.float32)
tab = gm.EmbeddingTable(emb); ds = gm.Dataset.ids(ub, it, uf, cf, y)
m = gm.DinNet(U, T, D, D, Cc).init_gaussian(np.random.default_rng(1))
m.set_embedding_training(0.5)
cfg = capi.default_train_cfg(batch=512, epochs=1, dropout_mode=0)
gm.train_steps(m, ds, cfg, 5, emb=tab)
capi.sync()
out = np.concatenate([m.get_weights(n).ravel() for n in ("mlp0", "mlp1", "mlp2", "att0")] + [tab.get_rows().ravel()])
np.save(%(out)r, out)
nb = m.sparse_exchange_bytes()
import os
if not %(comm)d:
    assert nb == 0
elif os.environ.get("GOCTR_EMB_FIXED_EXCHANGE", "1") == "0":
    # exact counts (two host read-backs per step): the last step's batch touches n ids: n x (4 + 8 D) bytes to the owner +
    # n x (4 + 4 D) gathered back (world 1: all to self)
    ids = np.concatenate([ub[0:512].ravel(), it[0:512]])           # (5 steps over 4 batches: the last one is batch 0 again)
    n = np.unique(ids[ids >= 0]).size
    assert nb == n * (4 + 8 * D) + n * (4 + 4 * D), (nb, n)
else:
    # fixed-size buckets: every bucket padded to S = the largest bucket of any batch (exact, from the plan), the owners' lists to
    # R = min(Vw, W S): the same bytes every step, known before the first one
    nk = [np.unique(np.concatenate([ub[k * 512:(k + 1) * 512].ravel(), it[k * 512:(k + 1) * 512]])) for k in range(4)]
    S = -(-max(int((u >= 0).sum()) for u in nk) // 4) * 4
    R = min(-(-V // 4) * 4, S)
    assert nb == S * (4 + 8 * D) + R * (4 + 4 * D), (nb, S, R)
if %(comm)d:
    capi.check(L.goctr_comm_destroy())
'''


def test_embedding_training_exchange_one_rank(tmp_path):
    """the bucketed sparse embedding exchange (counts all-gather, grouped ncclSend / ncclRecv of (id, fixed-point row) pairs to
    the owners, owner-side integer sums, all-gather of (id, delta); eager steps) with a one-rank communicator: every
    transfer is a self send and the sums are integer sums, so table and weights must equal the single-GPU graph-replayed
    run bit for bit; the bytes the exchange reports are exactly the touched ids' payload"""
    res = []
    # single GPU; one-rank RCCL with the fixed-size buckets (three captured graphs around the collectives, no host read-back);
    # the same eager (GOCTR_NO_GRAPH); the exact-count exchange with its two read-backs per step; and the atomics path's exchange
    for tag, comm, extra in (("single", 0, {}), ("fixed", 1, {}), ("fixed_eager", 1, {"GOCTR_NO_GRAPH": "1"}),
                             ("exact", 1, {"GOCTR_EMB_FIXED_EXCHANGE": "0"}), ("atomics", 1, {"GOCTR_EMB_PLAN": "0"})):
        out = str(tmp_path / f"emb_{tag}.npy")
        env = dict(os.environ, **extra)
        env["GOCTR_FORCE_COMM"] = str(comm)
        if tag == "atomics":
            env["GOCTR_EMB_FIXED_EXCHANGE"] = "0"
        r = subprocess.run([sys.executable, "-c", EMB_SCRIPT % dict(root=ROOT, comm=comm, out=out)], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        res.append(np.load(out))
    assert np.isfinite(res[0]).all()
    for k in (1, 2, 3):
        assert np.array_equal(res[0], res[k]), k                          # the plan path: identical integer sums whatever the exchange
    # (the atomics path associates the per-pair expression differently: float32 rounding apart, test_gpu_embtrain.py)
    assert np.max(np.abs(res[0] - res[4])) <= 2e-6 * max(1.0, float(np.max(np.abs(res[0]))))
