#!/usr/bin/env python3
"""N data-parallel training steps with trainable embeddings on a ONE-rank RCCL communicator (GOCTR_FORCE_COMM=1: every call of
the 8-GPU code path executes, every transfer is a self send).  Run twice under `rocprofv3 --hip-trace --stats` with N and 2N
steps: the difference of the two HIP API call tables is what N steps cost the host -- scripts/runs/*.sh asserts that it holds
no hipStreamSynchronize / hipMemcpy (the fixed-size sparse exchange has no host read-back), only graph launches."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["GOCTR_FORCE_COMM"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goctr_amd import capi, model as gm  # noqa: E402

steps = int(sys.argv[1])
capi.init(0)
L = capi.load()
idbuf = (C.c_uint8 * 128)()
capi.check(L.goctr_comm_unique_id(idbuf))
capi.check(L.goctr_comm_init(C.c_int(0), C.c_int(1), idbuf))
rng = np.random.default_rng(5)
rows, U, T, D, Cc, V, B = 1 << 15, 52, 50, 16, 53, 26744, 8192
emb = (rng.standard_normal((V, D)) * 0.25).astype(np.float32)
ub = ((rng.zipf(1.05, size=(rows, T)) - 1) % V).astype(np.int32)
it = rng.integers(0, V, size=rows).astype(np.int32)
uf = rng.random((rows, U), dtype=np.float32); cf = rng.random((rows, Cc), dtype=np.float32)
y = (rng.random(rows) < 0.5).astype(np.float32)
tab = gm.EmbeddingTable(emb); ds = gm.Dataset.ids(ub, it, uf, cf, y)
m = gm.DinNet(U, T, D, D, Cc)
r = np.random.default_rng(1)
for n in ("mlp0", "mlp1", "mlp2"):
    m.set_weights(n, (r.standard_normal(m._shape(n)) * 0.05).astype(np.float32))
m.set_embedding_training(0.05)
cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2, p0=0.005, p1=0.005, seed=3)
gm.train_steps(m, ds, cfg, 0, emb=tab)          # plan build + graph capture
gm.train_steps(m, ds, cfg, steps, emb=tab)      # ONE asynchronous call
capi.sync()
print("steps", steps, "exchange bytes per step", m.sparse_exchange_bytes())
capi.check(L.goctr_comm_destroy())
