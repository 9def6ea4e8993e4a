"""Where do the ~50 us per goctr_train_steps call go?  Host time of the (asynchronous) call vs time to completion, per call
length, DIN cfg3.   python scripts/launch_latency.py   (on the GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from goctr_amd import capi, model as gm


def main():
    capi.init(0)
    c = bench.CFG
    emb, ub, it, uf, cf, y = bench.synth(1 << 18, 42)
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, y)
    m = gm.DinNet(c["U"], c["T"], c["D"], c["D"], c["C"])
    bench.init_weights(m, 1, 1.0)
    cfg = capi.default_train_cfg(batch=c["B"], epochs=1, dropout_mode=2, p0=0.005, p1=0.005, seed=42)
    gm.train_steps(m, ds, cfg, 64, emb=tab)
    capi.sync()
    for n in (1, 2, 4, 8, 16, 20, 32, 64, 128):
        host, total = [], []
        for rep in range(7):
            capi.sync()
            t0 = time.perf_counter()
            gm.train_steps(m, ds, cfg, n, first_batch=rep, emb=tab)
            t1 = time.perf_counter()
            capi.sync()
            t2 = time.perf_counter()
            host.append((t1 - t0) * 1e6)
            total.append((t2 - t0) * 1e6)
        print(f"n={n:4d}: host call {min(host):7.1f} us, to completion {min(total):8.1f} us = {min(total) / n:6.1f} us/step", flush=True)
    # does the GPU slow down after idling?  20 steps after a pause of the given length (the bench's timed region follows
    # set-up work and 5 warm-up steps, not a hot loop)
    for pause_ms in (0, 1, 10, 100, 1000):
        total = []
        for rep in range(4):
            capi.sync()
            time.sleep(pause_ms * 1e-3)
            t0 = time.perf_counter()
            gm.train_steps(m, ds, cfg, 20, first_batch=rep, emb=tab)
            capi.sync()
            total.append((time.perf_counter() - t0) * 1e6 / 20)
        print(f"20 steps after {pause_ms:5d} ms idle: " + " ".join(f"{t:6.1f}" for t in total) + " us/step", flush=True)
    # ... and after 5 warm-up steps that follow the pause (what bench.py --warmup 5 --steps 20 does)
    for pause_ms in (100, 1000):
        total = []
        for rep in range(4):
            capi.sync()
            time.sleep(pause_ms * 1e-3)
            gm.train_steps(m, ds, cfg, 5, emb=tab)
            capi.sync()
            t0 = time.perf_counter()
            gm.train_steps(m, ds, cfg, 20, first_batch=5, emb=tab)
            capi.sync()
            total.append((time.perf_counter() - t0) * 1e6 / 20)
        print(f"5 warm-up + 20 steps after {pause_ms:5d} ms idle: " + " ".join(f"{t:6.1f}" for t in total) + " us/step", flush=True)
    # an empty call: the fixed host-side part
    ts = []
    for rep in range(7):
        capi.sync()
        t0 = time.perf_counter()
        capi.sync()
        ts.append((time.perf_counter() - t0) * 1e6)
    print(f"goctr_sync on an idle stream: {min(ts):.1f} us")


main()
