import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from goctr_amd import capi, model as gm
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
order = sys.argv[2] if len(sys.argv) > 2 else "AB"
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
for which in order:
    m = gm.DinNet(U, T, D, D, Cc).init_gaussian(np.random.default_rng(1))
    dev = W if which == "A" else 1
    for mode in (0, 2):
        c = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=mode, p0=0.01, p1=0.01, seed=7, devices=dev)
        for n in (1, 1, 2, 5):
            cost = gm.train_steps(m, ds, c, n, emb=tab, want_costs=True)
            capi.sync()
            w = flat(m)
            print(which, "devices", dev, "drop", mode, "steps", n, "cost", cost, "nan", int(np.isnan(w).sum()), flush=True)
            if which == "A":
                for k in range(1, W):
                    wk = flat(m.replica(k))
                    print("   replica", k, "equal", bool(np.array_equal(w, wk)), "nan", int(np.isnan(wk).sum()), "maxdiff", float(np.nanmax(np.abs(w - wk))), flush=True)
