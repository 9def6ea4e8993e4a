"""Step time of the DATA-PARALLEL code path on one GPU (a one-rank RCCL communicator, GOCTR_FORCE_COMM=1): graph a ->
ncclAllReduce -> graph b per step, against the single-GPU path (multi-step graphs, merged last launch) on the same box."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goctr_amd import capi, model as gm
import bench
comm = os.environ.get("GOCTR_FORCE_COMM", "0") == "1"
capi.init(0)
L = capi.load()
if comm:
    idbuf = (C.c_uint8 * 128)()
    capi.check(L.goctr_comm_unique_id(idbuf))
    capi.check(L.goctr_comm_init(C.c_int(0), C.c_int(1), idbuf))
emb, ub, it, uf, cf, y = bench.synth(1 << 18, 42)
tab = gm.EmbeddingTable(emb); ds = gm.Dataset.ids(ub, it, uf, cf, y)
m = gm.DinNet(52, 50, 16, 16, 53); bench.init_weights(m, 1)
cfg = capi.default_train_cfg(batch=8192, epochs=1, dropout_mode=2, p0=0.005, p1=0.005, seed=42)
gm.train_steps(m, ds, cfg, 40, emb=tab); capi.sync()
for K in (20, 200):
    ts = []
    for r in range(7):
        capi.sync(); t0 = time.perf_counter()
        gm.train_steps(m, ds, cfg, K, first_batch=(r * K) % 32, emb=tab); capi.sync()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"comm={int(comm)} K={K:4d}  median {ts[3]*1e6/K:7.2f} us/step")
