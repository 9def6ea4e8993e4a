"""usage: multi_dbg3.py W nmodels action...   action = <model idx>:<devices>:<steps>:<dropmode>"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from goctr_amd import capi, model as gm
W = int(sys.argv[1]); nm = int(sys.argv[2])
if W > 0: capi.init_devices([0] * W)
else: capi.init(0)
rng = np.random.default_rng(5)
rows, U, T, D, Cc, V = 4000, 52, 50, 16, 53, 500
emb = (rng.standard_normal((V, D)) * 0.25).astype(np.float32)
ub = rng.integers(-1, V, size=(rows, T)).astype(np.int32)
it = rng.integers(0, V, size=rows).astype(np.int32)
uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
y = (rng.random(rows) < 0.5).astype(np.float32)
tab = gm.EmbeddingTable(emb); ds = gm.Dataset.ids(ub, it, uf, cf, y)
names = ("mlp0", "mlp1", "mlp2", "att0")
def flat(m): return np.concatenate([m.get_weights(n).ravel() for n in names])
ms = [gm.DinNet(U, T, D, D, Cc).init_gaussian(np.random.default_rng(1)) for _ in range(nm)]
fb = [0] * nm
for act in sys.argv[3:]:
    i, dev, n, mode = map(int, act.split(":"))
    c = capi.default_train_cfg(batch=1024, epochs=1, dropout_mode=mode, p0=0.01, p1=0.01, seed=7, devices=dev)
    cost = gm.train_steps(ms[i], ds, c, n, first_batch=fb[i] % 4, emb=tab, want_costs=True)
    fb[i] += n
    capi.sync(); w = flat(ms[i])
    print(act, "cost", cost[:3], "nan", int(np.isnan(w).sum()), flush=True)
