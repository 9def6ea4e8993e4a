#!/usr/bin/env python3
"""driver of scripts/ubench/lds_hot_rows.hip: writes the bench's own id streams (frequency-ranked), builds and runs the
microbenchmark at cfg3 (V 26 744, D 16, B 8192) and cfg4 (V 10^7, D 64, B 16 384).  usage (GPU box): python scripts/lds_hot_rows.py"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "scripts", "ubench", "lds_hot_rows")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-Wno-unused-value", "-o", exe, os.path.join(ROOT, "scripts", "ubench", "lds_hot_rows.hip")], check=True)
for name, V, B, D, hots in (("cfg3", 26744, 8192, 16, (1024, 512, 256)), ("cfg4", 10_000_000, 16384, 64, (512, 256))):
    rng = np.random.default_rng(42)
    T = 50
    ub = (rng.zipf(1.05, size=(B, T)) - 1) % V
    it = (rng.zipf(1.05, size=B) - 1) % V
    ids = np.concatenate([ub, it[:, None]], 1)
    # frequency rank over a long stream of the same generator: hot rows = the lowest ids
    cnt = np.bincount(((rng.zipf(1.05, size=4_000_000) - 1) % V), minlength=V)
    rank = np.empty(V, np.int64); rank[np.argsort(-cnt, kind="stable")] = np.arange(V)
    ids = rank[ids]
    ids[np.concatenate([rng.random((B, T)) < 0.2, np.zeros((B, 1), bool)], 1)] = -1
    path = f"/tmp/ids_{name}.bin"
    ids.astype(np.int32).tofile(path)
    for hot in hots:
        print(name, end=": ", flush=True)
        subprocess.run([exe, path, str(B), str(T), str(V), str(D), str(hot)], check=False)
