"""cfg3 DIN step at 1 / 2 / 4 row tiles per CU (B = 8192 / 16384 / 32768) and half a chip (B = 4096): per-kernel durations under
rocprofv3 tell how much of each launch is per-tile work and how much is the launch's fixed cost (ramp, first operands, tail).
    rocprofv3 --kernel-trace --stats -d <dir> -- python scripts/tile_sweep.py <B>"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goctr_amd import capi, model as gm
B = int(sys.argv[1]); steps = 200
capi.init(0)
rng = np.random.default_rng(42)
U, T, D, Cc, V = 52, 50, 16, 53, 26744
rows = B * 4
p = 1.0 / np.arange(1, V + 1) ** 1.05; p /= p.sum()
ub = rng.choice(V, size=(rows, T), p=p).astype(np.int32); ub[rng.random((rows, T)) < 0.2] = -1
it = rng.choice(V, size=rows, p=p).astype(np.int32)
emb = (rng.standard_normal((V, D)) * 0.25).astype(np.float32)
uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
y = (rng.random(rows) < 0.5).astype(np.float32)
tab = gm.EmbeddingTable(emb); ds = gm.Dataset.ids(ub, it, uf, cf, y)
m = gm.DinNet(U, T, D, D, Cc).init_gaussian(np.random.default_rng(1))
cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2, p0=0.005, p1=0.005, seed=42)
gm.train_steps(m, ds, cfg, 20, emb=tab); capi.sync()
t0 = time.perf_counter()
gm.train_steps(m, ds, cfg, steps, first_batch=20, emb=tab); capi.sync()
dt = time.perf_counter() - t0
print(f"B {B}: {dt / steps * 1e6:.2f} us/step, {steps * B / dt / 1e6:.1f} M samples/s", flush=True)
