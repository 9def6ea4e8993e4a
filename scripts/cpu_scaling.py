"""how the CPU oracle (cpu_baseline leg of bench.py) scales with OpenMP threads on this box"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import pyoracle
c = bench.CFG
emb, ub, it, uf, cf, y = bench.synth(c["B"], 7)
X = pyoracle.assemble_rows(emb, ub, it, uf, cf)
for th in (1, 8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1):
        continue
    pyoracle.set_threads(th)
    m = pyoracle.CtrModel(pyoracle.DIN, c["U"], c["T"], c["D"], c["C"]).init_gaussian(np.random.default_rng(1))
    m.train(X, y, batch=c["B"], epochs=1)
    t0 = time.perf_counter(); m.train(X, y, batch=c["B"], epochs=2); dt = time.perf_counter() - t0
    print(f"threads {th:4d}: {2 * c['B'] / dt:10.0f} samples/s  ({dt / 2 * 1e3:.1f} ms/step)", flush=True)
