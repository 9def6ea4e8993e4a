import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from goctr_amd import capi, model as gm
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
capi.init_devices([0] * W)
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
B = 1024
mA = gm.DinNet(U, T, D, D, Cc).init_gaussian(np.random.default_rng(1))
mB = gm.DinNet(U, T, D, D, Cc).init_gaussian(np.random.default_rng(1))
cA = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2, p0=0.01, p1=0.01, seed=7, devices=W)
cB = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2, p0=0.01, p1=0.01, seed=7, devices=1)
def rep(tag, m, cost):
    capi.sync(); w = flat(m)
    print(tag, "cost", cost, "nan", int(np.isnan(w).sum()), flush=True)
rep("A13", mA, gm.train_steps(mA, ds, cA, 13, emb=tab, want_costs=True))
rep("A7", mA, gm.train_steps(mA, ds, cA, 7, first_batch=13 % 4, emb=tab, want_costs=True))
rep("B13", mB, gm.train_steps(mB, ds, cB, 13, emb=tab, want_costs=True))
rep("B7", mB, gm.train_steps(mB, ds, cB, 7, first_batch=13 % 4, emb=tab, want_costs=True))
for k in range(1, W):
    print("replica", k, bool(np.array_equal(flat(mA), flat(mA.replica(k)))))
print("maxdiff A vs B", float(np.max(np.abs(flat(mA) - flat(mB)))))
