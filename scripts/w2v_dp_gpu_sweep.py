"""item2vec across W = 8 logical ranks (loop-back communicator, one GPU): HS loss per path node of one cfg5 pass against the
exchange cadence and the per-rank parallelism.  (The oracle's 16-thread Hogwild run on the same corpus: 0.5587-0.5593.)
usage (GPU box): python scripts/w2v_dp_gpu_sweep.py > gpurun_out/w2v_dp_sweep.txt"""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from goctr_amd import capi, embedding as ge
from test_gpu_fullsize import _session_corpus, _hs_loss

W = int(os.environ.get("SWEEP_W", "8"))
capi.init_devices([0] * W)
rng = np.random.default_rng(105)
V, dim, n = 10681, 16, 10_000_000
doc, topics = _session_corpus(rng, V, n)
counts = np.bincount(doc, minlength=V) + 1
p0 = (rng.random((V, dim)) - 0.5) / dim
pos = rng.integers(1, n - 1, size=4000)
pairs = list(zip(doc[pos].tolist(), doc[pos + 1].tolist()))
paths = None
cases = [("every 1e5, min_pos 2048 (default)", 0, 2048), ("every 1e5, min_pos 256", 0, 256), ("every 2.5e4", 25000, 256), ("every 1e4", 10000, 256),
         ("every 1e5, min_pos 1024", 0, 1024), ("every 1e5, min_pos 4096", 0, 4096), ("every 2.5e4, min_pos 1024", 25000, 1024),
         ("once per pass", -1, 256), ("once per pass, min_pos 4096", -1, 4096)]
for name, every, min_pos in cases:
    os.environ["GOCTR_W2V_MIN_POS_DP"] = str(min_pos)
    capi.engine_select(0)
    m = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=False, streams=32768, slices=16, devices=W, exchange_every=every)
    m.create(counts, p0.copy())
    if paths is None:
        paths = m.get_paths()
    capi.sync(); t = time.perf_counter()
    m.train_pass(doc, doc.size, None, lr=0.025)
    capi.sync(); dt = time.perf_counter() - t
    loss = _hs_loss(m.get_param(), m.get_aux(), paths, pairs)
    print(json.dumps({"case": name, "W": W, "hs_loss": round(float(loss), 4), "pass_s": round(dt, 3)}), flush=True)
    m.close()
