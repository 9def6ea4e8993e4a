"""Fixed cost of one goctr_train_steps call at cfg3 (DIN, B 8192): wall time of K-step calls for several K, median of 15."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goctr_amd import capi, model as gm
import bench
emb, ub, it, uf, cf, y = bench.synth(1 << 16, 42)
tab = gm.EmbeddingTable(emb); ds = gm.Dataset.ids(ub, it, uf, cf, y)
m = gm.DinNet(52, 50, 16, 16, 53); bench.init_weights(m, 1)
cfg = capi.default_train_cfg(batch=8192, epochs=1)
gm.train_steps(m, ds, cfg, 40, emb=tab); capi.sync()
for K in (2, 4, 8, 16, 20, 32, 36, 48, 64, 128, 200):
    ts = []
    for r in range(15):
        capi.sync(); t0 = time.perf_counter()
        gm.train_steps(m, ds, cfg, K, first_batch=r % 4, emb=tab); capi.sync()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"K={K:4d}  median {ts[7]*1e6:9.1f} us  = {ts[7]*1e6/K:7.2f} us/step   min {ts[0]*1e6:9.1f}")
