#!/usr/bin/env python3
"""bench.py -- go-ctr hot path on MI355X: DIN training samples/sec (+ recommend QPS).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 either launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU; RANK / LOCAL_RANK /
WORLD_SIZE from the env) or, with no launcher around it, `python bench.py --gpus N` spawns its N ranks itself
(goctr_amd/launch.py).  The control plane (RCCL unique id, barrier, max over ranks) is a Unix-socket rendezvous:
no torch in the harness.
A "step" = one pass of the hot path (gather + attention + 3-layer MLP forward, BCE, backward, Adam)
over one batch of synthetic MovieLens-20M-shaped input, BASELINE.json config 3:
DIN cosine attention, T=50, D=16, U=52, C=53, batch 8192 per GPU, item vocab 26 744, ids + embedding
table resident in HBM before the timed region (id mode).  N>1 shards rows data-parallel ("weak":
per-GPU batch fixed) with one RCCL all-reduce of the flat gradient buffer per step.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     dominant kernel of the step, hipEvent-timed inside this process (an instrumented
               re-run of the same K steps, eager, event pair around every launch on the engine's
               stream); achieved = algorithmic flops (or bytes) per launch / average launch duration
  gather_roofline  the embedding gather + attention kernel against the HBM roof (north_star asks for it)
  cpu_baseline the CPU oracle (a port of the reference algorithm, oracle/) timed on this box's cores
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# BASELINE.json configs[2] / SURVEY.md section 8(d) cfg3
CFG = dict(U=52, T=50, D=16, C=53, H1=200, H2=80, V=26744, B=8192, PRED_B=4096, KIND="din")
FP64_MFMA_PEAK_TF = 78.6     # MI355X_MICROARCH.md: f64 MFMA peak
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: f32-input MFMA peak


def mlp_flops_per_sample(F: int, H: int) -> float:
    """sklearn-port MLP [F, H, 1], one training step, per sample (113 000 at F = 281, H = 100):
    forward 2 (F H + H); backward 2 F H (dW0) + 2 H (dW1) + 2 H (the hidden deltas) -- the first layer has no input gradient,
    so a step is TWO GEMMs of F x H, not three (rounds 2-5 priced three: VERDICT r5 weak 3)."""
    return 2.0 * (F * H + H) + 2.0 * (F * H + 2 * H)


def ctr_flops_per_sample(kind: str, U: int, T: int, D: int, C: int, H1: int = 200, H2: int = 80) -> float:
    """DIN / YouTube-DNN training step per sample, the dense part as this engine launches it (cfg3 MLP stage
    212 480, cfg4 282 880): forward I H1 + H1 H2 + H2, backward-data H2 + H2 H1 (+ H1 D for DIN's pooled segment), weight
    gradients I H1 + H1 H2 + H2 (+ T for att0), x 2; + 2 T D for the attention backward's dot products (DIN)."""
    I = U + 2 * D + C
    din = kind == "din"
    fwd = I * H1 + H1 * H2 + H2
    bwd = H2 + H2 * H1 + (H1 * D if din else 0)
    dw = I * H1 + H1 * H2 + H2 + (T if din else 0)
    return 2.0 * (fwd + bwd + dw) + (2.0 * T * D if din else 0.0)


def synth(rows: int, seed: int):
    """MovieLens-20M-shaped synthetic keys: Zipf(1.05) item ids, 20 % padded behaviour slots,
    U(0,1) dense side features, Bernoulli(0.5) labels."""
    rng = np.random.default_rng(seed)
    c = CFG
    ub = (rng.zipf(1.05, size=(rows, c["T"])) - 1) % c["V"]
    ub = ub.astype(np.int32)
    ub[rng.random((rows, c["T"])) < 0.2] = -1
    it = ((rng.zipf(1.05, size=rows) - 1) % c["V"]).astype(np.int32)
    mode = os.environ.get("GOCTR_BENCH_IDS", "")          # experiments only: uniform | none
    if mode == "uniform":
        ub = rng.integers(0, c["V"], size=(rows, c["T"])).astype(np.int32)
        it = rng.integers(0, c["V"], size=rows).astype(np.int32)
    elif mode == "none":
        ub[:] = -1
    uf = rng.random((rows, c["U"]), dtype=np.float32)
    cf = rng.random((rows, c["C"]), dtype=np.float32)
    y = (rng.random(rows) < 0.5).astype(np.float32)
    emb = rng.random((c["V"], c["D"]), dtype=np.float32) - 0.5 if c["V"] > 1_000_000 else \
        (rng.standard_normal((c["V"], c["D"])) * 0.25).astype(np.float32)
    return emb, ub, it, uf, cf, y


def init_weights(m, seed, scale=1.0):
    from goctr_amd import model as gm  # noqa: F401
    rng = np.random.default_rng(seed)
    I = CFG["U"] + 2 * CFG["D"] + CFG["C"]
    # random-init weights of the reference architecture (N(0,1) like din.go:187-191)
    m.set_weights("mlp0", (rng.standard_normal((I, CFG["H1"])) * scale).astype(np.float32))
    m.set_weights("mlp1", (rng.standard_normal((CFG["H1"], CFG["H2"])) * scale).astype(np.float32))
    m.set_weights("mlp2", (rng.standard_normal((CFG["H2"], 1)) * scale).astype(np.float32))


def attn_bwd_in_chain(train_emb=False):
    """DIN with frozen embeddings, D = 16, T <= 64: the chain launch ends with the att0 gradient's per-sample terms (what
    attn_bwd_kernel did in a launch of its own; csrc/ctr_chain_x3.h ChainX3Args::ab_*, GOCTR_CHAIN_ATTN_BWD=0 switches back)"""
    c = CFG
    return (c["KIND"] == "din" and not train_emb and c["D"] == 16 and c["T"] <= 64
            and os.environ.get("GOCTR_CHAIN_ATTN_BWD", "1") != "0")


def kernel_work(train_emb=False):
    """algorithmic work per launch of each kernel family for cfg3 (DESIGN.md 'Kernels')"""
    c = CFG
    B, I, H1, H2, T, D = c["B"], c["U"] + 2 * c["D"] + c["C"], c["H1"], c["H2"], c["T"], c["D"]
    ab_flops = 2.0 * B * T * D if attn_bwd_in_chain(train_emb) else 0.0
    din = c["KIND"] == "din"                             # (YouTube-DNN: no pooled-segment gradient product, no att0 column -- rounds 2-5
    dpD, attT = (D if din else 0), (T if din else 0)     #  priced both into its chain / dW0 launches: its chain frac read ~14 % high)
    gather_bytes = B * ((T + 1) * D * 4 + (T + 1) * 4)   # 3 264 B rows + 204 B ids per sample
    return {
        "attn_fwd": ("hbm", gather_bytes), "attn_bwd": ("hbm", gather_bytes),
        "gemm_fwd0": ("mfma", 2.0 * B * I * H1), "gemm_fwd1": ("mfma", 2.0 * B * H1 * H2),
        "gemm_out": ("mfma", 2.0 * B * H2), "bwd_dz1": ("mfma", 2.0 * B * H2),
        "bwd_dz0": ("mfma", 2.0 * B * H2 * H1), "bwd_dp": ("mfma", 2.0 * B * H1 * D),
        # since the fused kernels: "dW0" = ONE launch computing dW0 + dW1 + dW2 + datt0,
        # "chain" = layers 0..2 forward + BCE + dz1 + dz0 + dp per 32-row tile
        "dW0": ("mfma", 2.0 * B * (I * H1 + H1 * H2 + H2 + attT)),
        "dW1": ("mfma", 2.0 * B * H1 * H2), "dW2": ("mfma", 2.0 * B * H2),
        # (+ the T x D dot products of the attention backward where the chain launch ends with them)
        "chain": ("mfma", 2.0 * B * (I * H1 + H1 * H2 + H2 + H2 + H2 * H1 + H1 * dpD) + ab_flops),
    }


def chain_algorithmic_bytes(train_emb=False):
    """HBM bytes the fused chain launch must move per training launch: h0 in; A0, dz0, dz1, yhat, loss terms out (until round 5 also
    A1, dz2 and, DIN, dp) -- padded widths, as stored.  Where the launch ends with the
    attention backward: + the behaviour ids, gates and similarity weights in, the per-sample att0 terms out, and every
    distinct table row once (the rows themselves are gathered B x T times, from L2 at cfg3's 1.7 MB table)."""
    c = CFG
    I = c["U"] + 2 * c["D"] + c["C"]
    Ip, H1p, H2p = -(-I // 16) * 16, 208, 80
    Dp = -(-c["D"] // 16) * 16 if c["KIND"] == "din" else 0
    Tp = -(-c["T"] // 16) * 16
    ab = attn_bwd_in_chain(train_emb)
    # round 6 (GOCTR_CHAIN_TILE_SUMS, default): dW2 and the att0 terms leave as per-tile sums ([B / 32] x (H2p + Tp) floats) instead of
    # their operands A1 [B, H2p], dz2 [B, 16], the per-sample terms [B, Tp] -- and dp [B, Dp], whose only reader was the launch's own tail
    sums = os.environ.get("GOCTR_CHAIN_TILE_SUMS", "1") != "0"
    if sums:
        n = 4 * c["B"] * (Ip + 2 * H1p + H2p + 3) + 4 * (c["B"] // 32) * (H2p + (Tp if ab else 0)) + (0 if ab else 4 * c["B"] * Dp)
    else:
        n = 4 * c["B"] * (Ip + Dp + 2 * H1p + 2 * H2p + 16 + 3)
    if ab:
        n += 4 * c["B"] * (3 * c["T"] + (0 if sums else Tp)) + 4 * c["D"] * min(c["B"] * c["T"], c["V"])
    return n


def roofline_obj(kind, work, avg_ms):
    if kind == "hbm":
        ach = work / (avg_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None}
    ach = work / (avg_ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(ach, 3), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": round(ach / FP32_MFMA_PEAK_TF, 4), "traffic": None}


L2_PEAK_GBS = 34500.0        # MI355X_MICROARCH.md "L2": ~34.5 TB/s aggregate


def pmc_entry(workload, phase, symbol, grid_threads=None, need_bytes=True):
    """The rocprofv3 summary of kernel `symbol` (name + template arguments, as goctr_prof_kernel reports it) in `phase`
    (train / predict) of `workload` (din / youtube / dinemb / youtubeemb / mlp / item2vec / knn): memory-side bytes per
    launch (separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x 2 as MI355X_MICROARCH.md prescribes for gfx950), L2 hit
    rate, SQ counters, rocprofv3's own average duration -- from the newest committed profiles/*_<workload>_kernels.json,
    which scripts/prof_workload.sh + scripts/prof_summarize.py wrote from THIS command, one process per pass and phase.
    The counters cannot be collected inside the benchmark run itself (they need rocprofv3 around the process), so the file
    names its commit.  {} when no summary is committed or it holds no entry for exactly this kernel symbol (a summary of
    another kernel is never substituted)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_kernels.json")))
    if not files or not symbol:
        return {}
    try:
        d = json.load(open(files[-1]))
        # ("name*": any template instantiation of `name` -- for the single-variant kernels of the mlp / item2vec / knn engines)
        match = (lambda k: k.startswith(symbol[:-1])) if symbol.endswith("*") else (lambda k: k == symbol)
        cands = [e for e in d["phases"].get(phase, {}).values() if match(e.get("kernel", "")) and (e.get("hbm_bytes") is not None or not need_bytes)]
        if grid_threads is not None:
            cands = [e for e in cands if e.get("grid_threads") == grid_threads]
        if not cands:
            return {}
        t = dict(max(cands, key=lambda e: e.get("calls", 0)))
        t["source"] = os.path.basename(files[-1])
        t["commit"] = d.get("commit")
        t["phase"] = phase
        return t
    except Exception:
        return {}


def with_traffic(rl, workload, phase, symbol, grid_threads=None, avg_ms=None, algorithmic_bytes=None):
    """fill roofline.traffic (+ provenance, L2 hit rate, MFMA-busy) from the committed rocprofv3 summary of exactly this
    kernel symbol and launch shape; with avg_ms also the memory-side rate that traffic means at the duration measured live
    in this run.  An entry whose traffic is below half the algorithmic bytes is refused (it cannot be this launch)."""
    t = pmc_entry(workload, phase, symbol, grid_threads)
    rl["kernel_symbol"] = t.get("kernel", symbol)
    if not t:
        rl["traffic_note"] = f"no committed rocprofv3 summary for {phase}/{symbol}" + (f"@{grid_threads}" if grid_threads else "")
        return rl
    if algorithmic_bytes and t["hbm_bytes"] < 0.5 * algorithmic_bytes:
        rl["traffic_note"] = (f"committed summary for {symbol} reports {t['hbm_bytes']} B < half the algorithmic {algorithmic_bytes} B: "
                              "refused (not this launch)")
        return rl
    rl["traffic"] = round(t["hbm_bytes"])
    rl["traffic_source"], rl["traffic_commit"], rl["traffic_phase"] = t["source"], t["commit"], phase
    rl["traffic_grid_threads"] = t.get("grid_threads")
    if t.get("l2_hit_rate") is not None:
        rl["l2_hit_rate"] = t["l2_hit_rate"]
    if t.get("avg_us"):
        rl["avg_us_rocprofv3"] = t["avg_us"]
    if (t.get("sq") or {}).get("mfma_busy_pct") is not None:
        rl["mfma_busy_pct"] = t["sq"]["mfma_busy_pct"]
    if avg_ms:
        rl["hbm_side_GBs"] = round(t["hbm_bytes"] / (avg_ms * 1e-3) / 1e9, 1)
    return rl


def usable_cores() -> int:
    """host cores this process may actually use: min(affinity mask, cgroup CPU quota).  The GPU boxes run the
    job in a container with a CPU quota far below the 256 logical CPUs nproc reports; oversubscribing it
    collapses OpenMP throughput (measured: 686 k samples/s at 32 threads vs 16 k at 256)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())             # cgroup v1
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // p)))
        except Exception:
            pass
    return n


def cpu_baseline(budget_s=15.0):
    """the CPU oracle (port of the reference algorithm, float32, OpenMP over rows on all host cores)
    timed on a bounded sample of the same workload: dense-X DIN training steps at B=8192."""
    from oracle import pyoracle
    c = CFG
    cores = usable_cores()
    pyoracle.set_threads(cores)
    rows = c["B"]
    emb, ub, it, uf, cf, y = synth(rows, 7)
    X = pyoracle.assemble_rows(emb, ub, it, uf, cf)
    kind = pyoracle.YOUTUBE if c["KIND"] == "youtube" else pyoracle.DIN
    m = pyoracle.CtrModel(kind, c["U"], c["T"], c["D"], c["C"]).init_gaussian(np.random.default_rng(1))
    t0 = time.perf_counter()
    pd = 0.003 if c["KIND"] == "youtube" else 0.005
    m.train(X, y, batch=c["B"], epochs=1, drop_mode=2, p0=pd, p1=pd, seed=42)          # one step (also warms the caches)
    one = time.perf_counter() - t0
    steps = int(max(2, min(200, budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    m.train(X, y, batch=c["B"], epochs=steps, drop_mode=2, p0=pd, p1=pd, seed=42)      # rows == batch => epochs == steps
    dt = time.perf_counter() - t0
    return {"value": round(steps * rows / dt, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{steps} {c['KIND']} training steps at batch {rows} (dense TrainSample rows, T=50, D={c['D']}), "
                      f"oracle/orc_ctr.c with {cores} OpenMP threads (= the container's CPU quota; "
                      f"{os.cpu_count()} logical CPUs visible), {dt:.1f} s"}


def serving_qps(m, tab, emb, ub, it, uf, cf, n=1 << 16, reps=5):
    """recommend QPS through the drop-in boundary as a Go host would drive it ("with and without host row
    assembly"), PCIe-inclusive, rank 0 only:
      recommend_qps_keys       recommend.BatchPredict on the device (goctr_batch_predict): n sample keys (16 B each) in HOST
                               memory -> behaviour-cache lookup + row assembly + predict on the GPU -> n scores back
      recommend_qps_host_rows  model.Predict's own convention (goctr_predict_dense): n dense [XCols] float32 rows in HOST
                               memory (assembled beforehand, untimed, by the product's own gather) -> scores back"""
    import ctypes as C
    from goctr_amd import capi, model as gm, recommend as gr
    L = capi.load()
    c = CFG
    rng = np.random.default_rng(5)
    res = {}
    # --- keys: a behaviour cache of 8192 users (histories of 20..120 items, newest first), feature tables for every user / item
    n_users, V = 8192, c["V"]
    lens = rng.integers(20, 121, size=n_users)
    off = np.zeros(n_users + 1, np.int64)
    off[1:] = np.cumsum(lens)
    seq_items = ((rng.zipf(1.05, size=int(off[-1])) - 1) % V).astype(np.int32)
    seq_ts = np.concatenate([np.sort(rng.integers(1, 1 << 30, size=k))[::-1] for k in lens]).astype(np.int64)
    ut = rng.random((n_users, c["U"]), dtype=np.float32)
    itab = rng.random((V, c["C"]), dtype=np.float32)
    ubc, rs = C.c_void_p(), C.c_void_p()
    capi.check(L.goctr_ubcache_create(C.c_int64(n_users), capi.ptr(off, C.c_int64), capi.ptr(seq_items, C.c_int32),
                                      capi.ptr(seq_ts, C.c_int64), C.byref(ubc)))
    capi.check(L.goctr_recsys_create(ubc, tab._h, capi.ptr(ut, C.c_float), C.c_int64(n_users), C.c_int(c["U"]),
                                     capi.ptr(itab, C.c_float), C.c_int64(V), C.c_int(c["C"]), C.byref(rs)))
    users = rng.integers(0, n_users, size=n).astype(np.int32)
    items = ((rng.zipf(1.05, size=n) - 1) % V).astype(np.int32)
    ts = rng.integers(1, 1 << 30, size=n).astype(np.int64)
    scores = np.empty(n, np.float32)

    def call():
        capi.check(L.goctr_batch_predict(m._h, rs, capi.ptr(users, C.c_int32), capi.ptr(items, C.c_int32), capi.ptr(ts, C.c_int64),
                                         C.c_int64(n), C.c_int(c["PRED_B"]), capi.ptr(scores, C.c_float), None, None))
    call()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    dtk = (time.perf_counter() - t0) / reps
    res["recommend_qps_keys"] = round(n / dtk, 1)
    res["keys_note"] = (f"goctr_batch_predict: {n} (user, item, ts) keys = {n * 16 / 1e6:.1f} MB over PCIe per call, device-side "
                        f"ubcache lookup + row assembly + predict at batch {c['PRED_B']}, scores back: {dtk * 1e3:.2f} ms per call")
    L.goctr_recsys_destroy(rs)
    L.goctr_ubcache_destroy(ubc)
    # --- dense rows in host memory
    X = tab.gather_rows(ub[:n], it[:n], uf[:n], cf[:n])      # (the product's own gather, goctr_gather_rows; untimed)
    si = gr.SampleInfo.from_dims(c["U"], c["T"], c["D"], c["C"])
    gm.Predict(m, X.shape[0], c["PRED_B"], si, X)
    t0 = time.perf_counter()
    for _ in range(3):
        gm.Predict(m, X.shape[0], c["PRED_B"], si, X)
    dth = (time.perf_counter() - t0) / 3
    res["recommend_qps_host_rows"] = round(X.shape[0] / dth, 1)
    res["host_rows_note"] = (f"goctr_predict_dense: {X.shape[0]} dense rows x {X.shape[1]} f32 = {X.nbytes / 1e6:.0f} MB over PCIe per call, "
                             f"predict, y back: {dth * 1e3:.1f} ms per call")
    return res


def rank_serving(kind="din", seconds=0.3):
    """recommend.Rank at the request size the reference's HTTP API sees (recommend/api.go:106-131: one user, a short
    itemIdList), from 1 and 8 concurrent host threads: goctr_amd/host/rank_bench (C++ above the C-ABI, std::thread standing in
    for the handler goroutines; Python threads would serialise on the GIL).  Returns the fields merged into the bench line."""
    import subprocess
    exe = os.path.join(ROOT, "goctr_amd", "host", "rank_bench")
    if not os.path.exists(exe):
        return {"rank_note": "goctr_amd/host/rank_bench not built"}
    try:
        r = subprocess.run([exe, "--threads", "1,8", "--n", "32,256,2048", "--seconds", str(seconds), "--kind", kind, "--coalesce", "both"],
                           capture_output=True, text=True, timeout=120)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                       # noqa: BLE001
        return {"rank_note": f"rank_bench failed: {e}"}
    lat, qps, lat_nc, qps_nc = {}, {}, {}, {}
    for e in d["results"]:
        key = f"n{e['n']}_t{e['threads']}"
        (lat if e["coalesce"] else lat_nc)[key] = e["latency_us"]["p50"]
        (qps if e["coalesce"] else qps_nc)[key] = round(e["rank_qps"], 1)
    return {"rank_latency_us": lat, "rank_qps": qps, "rank_latency_us_no_coalescing": lat_nc, "rank_qps_no_coalescing": qps_nc,
            "rank_p99_us": {f"n{e['n']}_t{e['threads']}": e["latency_us"]["p99"] for e in d["results"] if e["coalesce"]},
            "rank_bit_equal_to_single_threaded": d.get("bit_equal_to_single_threaded"),
            "rank_note": d["workload"] + "; p50 latency per call / calls per second; key n<candidates>_t<host threads>; serving slots with own "
                         "streams + pinned staging; calls of <= 1024 rows that arrive while a pass is in flight are coalesced into one pass "
                         "(GOCTR_SERVE_COALESCE=0 for the *_no_coalescing figures)"}


def _emit(out):
    print(json.dumps(out))


def bench_mlp(args):
    """BASELINE configs[1] / SURVEY 8(d) cfg2: sklearn-port MLP [281,100,1] relu/adam (float64), B = 4096, rows resident
    in HBM; a step = forward, log-loss, backward, per-parameter Adam over one batch (nn/neural_network/basemlp64.go)."""
    from goctr_amd import capi, mlp as gmlp
    capi.init(0)
    F, H, B, rows = 281, 100, 4096, 1 << 20
    rng = np.random.default_rng(42)
    X = rng.random((rows, F), dtype=np.float32)
    y = (rng.random(rows) < 0.5).astype(np.float32)
    clf = gmlp.MLPClassifier([H], "relu", "adam", 1e-5)
    clf.BatchSize = B
    units = [F, H, 1]
    clf.create(units, B, clf.init_params(units, rng))
    clf.upload(X, y)
    # at least 1200 warm-up steps (~50 ms): the GPU idled while the rows above were generated and runs ~6 % slower for its
    # next milliseconds; the line reports the warm-up it actually did
    # (~160 ms: the clocks need ~100 ms of load, see main().  Not in a profiling pass -- scripts/prof_workload.sh runs --phase train:
    # 4000 eager steps under rocprofv3 --pmc take minutes, round 6 lost an hour to passes that were merely that slow)
    warm = max(args.warmup, 4000) if args.phase == "all" else args.warmup
    clf.train_steps(warm)
    capi.sync()
    regions = []                                           # the median of --regions back-to-back regions of exactly --steps steps
    for r in range(max(args.regions, 1)):
        capi.sync()
        t0 = time.perf_counter()
        clf.train_steps(args.steps, first_batch=warm + r * args.steps)
        capi.sync()
        regions.append(time.perf_counter() - t0)
    dt = sorted(regions)[(len(regions) - 1) // 2]
    flops = B * mlp_flops_per_sample(F, H)                 # 113 000 per sample (two F x H GEMMs: no input gradient)
    out = {"metric": "training samples/sec (sklearn-port MLP [281,100,1], float64)", "value": round(args.steps * B / dt, 1),
           "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": warm,
           "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic", "timed_regions": len(regions), "timed_regions_ms": [round(x * 1e3, 4) for x in regions],
           "config": {"workload": "BASELINE configs[1]: MLP [281,100,1] relu/adam alpha=1e-5, batch 4096, 2^20 rows resident in HBM",
                      "global_batch": B, "parallelism": "dp1"},
           "roofline": {"bound": "mfma", "achieved": round(flops / (dt / args.steps) / 1e12, 3), "peak": FP64_MFMA_PEAK_TF,
                        "unit": "TFLOP/s", "frac": round(flops / (dt / args.steps) / 1e12 / FP64_MFMA_PEAK_TF, 4),
                        "traffic": None, "flops_per_sample": mlp_flops_per_sample(F, H),
                        "kernel": "whole step (3 launches; latency-bound at this size: see dominant_kernel)"}}
    # memory-side bytes of the step's three launches (PMC summary of this command), and the longest kernel on its own
    names = ("mlp_chain_kernel", "mlp_tn64_kernel", "mlp_reduce_update_kernel")
    per = {k: pmc_entry("mlp", "train", k + "*") for k in names}
    if all(per.values()):
        out["roofline"]["traffic"] = round(sum(v["hbm_bytes"] for v in per.values()))
        out["roofline"]["traffic_source"] = per[names[0]]["source"]
        out["roofline"]["traffic_commit"] = per[names[0]]["commit"]
        out["kernels_rocprofv3_us"] = {v["kernel"]: v.get("avg_us") for k, v in per.items()}
        # both GEMM-carrying kernels do 2 B (F+1) H flops (the chain kernel: the hidden layer; tn64: its weight gradient)
        gf = 2.0 * B * (F + 1) * H
        dom = max(("mlp_chain_kernel", "mlp_tn64_kernel"), key=lambda k: per[k].get("avg_us") or 0.0)
        t = per[dom].get("avg_us")
        out["dominant_kernel"] = {"kernel": dom + (" (gather + hidden layer + output unit + deltas, f64 MFMA)" if dom == "mlp_chain_kernel"
                                                   else " (weight-gradient GEMM, f64 MFMA)"),
                                  "kernel_symbol": per[dom]["kernel"], "flops": gf, "avg_us_rocprofv3": t, "traffic": round(per[dom]["hbm_bytes"]),
                                  "mfma_busy_pct": (per[dom].get("sq") or {}).get("mfma_busy_pct"),
                                  "frac_of_f64_mfma_peak": round(gf / (t * 1e-6) / 1e12 / FP64_MFMA_PEAK_TF, 4) if t else None}
    if not args.no_cpu_baseline:
        from oracle import pyoracle
        cfg = pyoracle.mlp_cfg(units, "relu", alpha=1e-5)
        theta = clf.init_params(units, np.random.default_rng(1))
        opt = pyoracle.MlpOptimizer("adam", theta.size)
        cores = usable_cores()
        pyoracle.set_threads(cores)                       # OpenMP over rows / parameter rows, all quota cores
        n = 16 * B
        Xd, yd = X[:n].astype(np.float64), y[:n, None].astype(np.float64)
        pm = np.stack([np.arange(n)] * 8).astype(np.int32)
        pyoracle.mlp_fit(cfg, theta.copy(), pyoracle.MlpOptimizer("adam", theta.size), Xd, yd, B, 1, tol=-1.0, perm=pm[:1])   # warm-up epoch
        t0 = time.perf_counter()
        pyoracle.mlp_fit(cfg, theta, opt, Xd, yd, B, 8, tol=-1.0, perm=pm)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(8 * n / dtc, 1), "unit": "samples/s", "cores": cores, "kind": "port",
                               "sample": f"8 epochs over {n} rows at batch {B} after 1 warm-up epoch, oracle/orc_sklmlp.c (float64 port of "
                                         f"basemlp64.go, OpenMP over rows on {cores} threads = the container's CPU quota), {dtc:.1f} s"}
    _emit(out)


def bench_mlp100k(args):
    """BASELINE configs[0]: the reference's own end-to-end run (main.go:39-50, README.md:160-165 "28 s"): the 2-layer MLP
    [281, 100, 1] relu / adam, alpha 1e-5, on MovieLens-100k's 79 948 training rows x 281 features, BatchSize 200, 20 epochs --
    399 whole batches and ONE short batch of 148 rows per epoch (quirk Q11), 8 000 updates in all, through goctr_mlp_fit's
    resident part (goctr_mlp_fit_resident: rows in HBM when the timed region starts; the shuffle order of every epoch comes
    from the host like the reference's in-place shuffle and is uploaded per epoch inside the timed region).
    A "step" here is one EPOCH (400 updates); value = training samples/s over the 20 epochs; `fit_wall_s` stands next to the
    README's 28 s (context only: another machine, the Go CPU path).  The shape is pure launch latency: a 200-row batch is 13
    workgroups of a 256-CU chip."""
    from goctr_amd import capi, mlp as gmlp
    capi.init(0)
    n, F, H, B, iters = 79948, 281, 100, 200, 20
    rng = np.random.default_rng(22)
    X = rng.random((n, F), dtype=np.float32)
    y = ((X[:, :8].sum(1) + 0.3 * rng.standard_normal(n)) > 4).astype(np.float32).reshape(-1, 1)
    units = [F, H, 1]
    clf = gmlp.MLPClassifier([H], "relu", "adam", 1e-5)
    clf.BatchSize, clf.MaxIter, clf.Tol = B, iters, -1.0          # (Tol < 0: all 20 epochs, like the reference's run which does not converge earlier)
    theta0 = clf.init_params(units, rng)
    order, perms = np.arange(n), []
    for _ in range(iters):                                         # cumulative in-place shuffles (fitStochastic, basemlp64.go:786-788)
        order = order[rng.permutation(n)]
        perms.append(order.copy())
    perm = np.stack(perms).astype(np.int32)
    clf.create(units, B, theta0)
    t0 = time.perf_counter()
    clf.upload(X, y)
    capi.sync()
    upload_s = time.perf_counter() - t0
    clf.FitResident(perm)                                          # untimed first fit from the fresh state: graph capture, clocks
    first_curve = list(clf.LossCurve)                              # (the timed fits below continue from it: same kernels, same 8 000 updates each)
    regions = []
    for _ in range(max(min(args.regions, 5), 1)):
        capi.sync()
        t0 = time.perf_counter()
        clf.FitResident(perm)
        capi.sync()
        regions.append(time.perf_counter() - t0)
    dt = sorted(regions)[(len(regions) - 1) // 2]
    nb = -(-n // B)
    updates = iters * nb
    us_per_update = dt / updates * 1e6
    flops = iters * n * mlp_flops_per_sample(F, H)
    # what the launches alone cost: three per whole-batch update (mlp_chain, mlp_tn64, mlp_reduce_update) at the guide's
    # dependent-kernel boundary (MI355X_MICROARCH.md price list, "boundary": 1.1-1.9 us inside a replayed graph) + the MFMA
    # time of a 200-row batch on the 13 CUs it occupies
    launches = 3
    boundary_us = 1.5
    mfma_us = B * mlp_flops_per_sample(F, H) / (FP64_MFMA_PEAK_TF * 1e12 * 13 / 256) * 1e6
    floor_us = launches * boundary_us + mfma_us
    out = {"metric": "training samples/sec (sklearn-port MLP [281,100,1], float64, the reference's own run)",
           "value": round(iters * n / dt, 1), "unit": "samples/s", "n_gpus": 1, "steps": iters, "warmup": iters,
           "ms_per_step": round(dt / iters * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "timed_regions": len(regions), "timed_regions_ms": [round(x * 1e3, 3) for x in regions],
           "config": {"workload": "BASELINE configs[0]: MLP [281,100,1] relu/adam alpha=1e-5 on 79 948 x 281 MovieLens-100k-shaped rows, "
                                  "batch 200, 20 epochs (399 whole batches + one of 148 rows per epoch), rows resident in HBM; a step = one epoch",
                      "global_batch": B, "parallelism": "dp1"},
           "fit_wall_s": round(dt, 4), "updates": updates, "us_per_update": round(us_per_update, 2),
           "upload_s_untimed": round(upload_s, 4),
           "fit_wall_s_incl_upload": round(dt + upload_s, 4),
           "reference_readme_s": 28.0,
           "reference_note": "README.md:160-165 quotes 28 s for this run on the author's CPU (Go, gonum): context, not a baseline measured here",
           "loss_first_and_last_epoch_of_the_first_fit": [first_curve[0], first_curve[-1]] if first_curve else None,
           "roofline": {"bound": "mfma", "achieved": round(flops / dt / 1e12, 4), "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(flops / dt / 1e12 / FP64_MFMA_PEAK_TF, 5), "traffic": None,
                        "kernel": "whole fit (three launches per update on 13 of 256 CUs: launch-latency-bound by construction)",
                        "launch_floor": {"launches_per_update": launches, "boundary_us_each": boundary_us, "mfma_us_per_update": round(mfma_us, 3),
                                         "floor_us_per_update": round(floor_us, 2), "measured_us_per_update": round(us_per_update, 2),
                                         "measured_over_floor": round(us_per_update / floor_us, 2)}}}
    names = ("mlp_chain_kernel", "mlp_tn64_kernel", "mlp_reduce_update_kernel")
    per = {k: pmc_entry("mlp100k", "train", k + "*", need_bytes=False) for k in names}      # (kernel trace only: no counter passes for this line)
    if all(per.values()):
        out["kernels_rocprofv3_us"] = {v["kernel"]: v.get("avg_us") for v in per.values()}
        out["kernels_rocprofv3_source"] = per[names[0]]["source"]
        ksum = sum(v.get("avg_us") or 0.0 for v in per.values())
        out["roofline"]["launch_floor"]["sum_of_kernel_us_rocprofv3"] = round(ksum, 2)
        out["roofline"]["launch_floor"]["longest_launch"] = max(per.values(), key=lambda v: v.get("avg_us") or 0.0)["kernel"]
    if not args.no_cpu_baseline:
        from oracle import pyoracle
        cores = usable_cores()
        pyoracle.set_threads(cores)
        cfg = pyoracle.mlp_cfg(units, "relu", alpha=1e-5)
        theta = theta0.copy()
        opt = pyoracle.MlpOptimizer("adam", theta.size)
        Xd, yd = X.astype(np.float64), y.astype(np.float64)
        ne = 4                                                   # a bounded sample: 4 of the 20 epochs (1 600 updates)
        t0 = time.perf_counter()
        ref = pyoracle.mlp_fit(cfg, theta, opt, Xd, yd, B, ne, tol=-1.0, perm=perm[:ne])
        dtc = time.perf_counter() - t0
        pyoracle.set_threads(1)
        out["cpu_baseline"] = {"value": round(ne * n / dtc, 1), "unit": "samples/s", "cores": cores, "kind": "port",
                               "sample": f"the first {ne} of the 20 epochs of the same run (same rows, init and shuffles), oracle/orc_sklmlp.c "
                                         f"(float64 port of basemlp64.go, OpenMP over rows on {cores} threads = the container's CPU quota), "
                                         f"{dtc:.1f} s => {dtc / ne * iters:.1f} s for 20 epochs",
                               "loss_after_sample_epochs": float(ref[-1]), "device_loss_at_same_epoch": first_curve[ne - 1]}
    _emit(out)


def bench_item2vec(args):
    """BASELINE configs[4] / SURVEY 8(d) cfg5: SkipGram + hierarchical softmax, window 5, D = 16 float64, V = 10 681,
    Zipf(1.0) 10^7-word corpus resident in HBM; a step = one pass over the corpus (Hogwild kernel: the doc is cut into 16
    slices like the reference cuts it over runtime.NumCPU() goroutines, options.go:41 -- windows are clipped at those
    16 ends only -- and every slice is walked by 2048 lane groups, 32768 workers in all; skip-gram + HS runs the node-major kernel
    w2v_hogwild_nm_kernel: the pairs of a position walk the centre word's path four at a time, a node is read and updated once
    per chunk instead of once per pair)."""
    from goctr_amd import capi, embedding as ge
    capi.init(0)
    V, dim, n, streams = 10681, 16, 10_000_000, 32768
    rng = np.random.default_rng(42)
    p = 1.0 / np.arange(1, V + 1)
    p /= p.sum()
    doc = rng.choice(V, size=n, p=p).astype(np.int32)
    counts = np.bincount(doc, minlength=V) + 1
    m = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=False, streams=streams, slices=16)
    m.create(counts)
    m.upload_doc(doc)
    steps, warm = max(1, args.steps // 20), max(1, args.warmup // 20)
    for _ in range(warm):
        m.train_resident(n * (steps + warm), lr=0.025)
    capi.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.train_resident(n * (steps + warm), lr=0.025)
    capi.sync()
    dt = time.perf_counter() - t0
    wps = steps * n / dt
    # ---- roofline of the IMPLEMENTED walk (rounds 1-4 priced the reference's pair-major row traffic, 19 968 B
    # per word, which this kernel no longer performs -- the fraction read 1.03).  w2v_hogwild_nm_kernel per corpus position: the
    # centre word's Huffman path (L nodes) is walked once per CHUNK of up to JB = 4 context words; a node visit reads the node
    # vector (dim x 8 B) and adds one update to it, a context word is read once and updated once.  Rows in the LDS hot tables
    # (the 256 most frequent words and the 256 heaviest nodes at dim 16) cost no memory traffic between merges; COLD rows are
    # device-scope loads and atomic adds, which the eight non-coherent L2s pass through to the fabric.  From this corpus' own
    # paths and counts (window shrink uniform in 0 .. 4 => 2 (5 - shrink) contexts, ceil(. / 4) chunks: 6 and 1.8 on average):
    off, nodes, _codes = m.get_paths()
    off = np.asarray(off, np.int64); nodes = np.asarray(nodes, np.int64)
    hot_rows = 256                                            # csrc/w2v.hip run_pass: rows_cached at WPS 4, dim 16 (GOCTR_W2V_HOT=1)
    plen = np.diff(off)
    node0 = (V - 1) - min(hot_rows, V - 1)
    cold_nodes = np.add.reduceat((nodes < node0).astype(np.int64), off[:-1].clip(max=max(nodes.size - 1, 0))) * (plen > 0)
    hot_word = np.zeros(V, bool); hot_word[np.argsort(-counts, kind="stable")[:hot_rows]] = True
    freq = np.bincount(doc, minlength=V) / float(n)
    row = dim * 8
    chunks, ctxs = 1.8, 6.0
    L_mean, Lc_mean = float((freq * plen).sum()), float((freq * cold_nodes).sum())
    p_cold_ctx = float(freq[~hot_word].sum())
    fabric_b = chunks * Lc_mean * 2 * row + ctxs * p_cold_ctx * 2 * row + 4          # cold rows + the doc id
    all_b = chunks * L_mean * 2 * row + ctxs * 2 * row + 4                            # every row the walk touches (LDS + fabric)
    out = {"metric": "item2vec training words/sec (SkipGram + HS, float64)", "value": round(wps, 1), "unit": "words/s",
           "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "BASELINE configs[4]: SkipGram+HS, window 5, D=16, V=10681, Zipf(1.0), 10^7-word corpus "
                                  "resident in HBM, one pass per step, Hogwild: 16 slices (window clipping as in the reference) x "
                                  "2048 workers, node-major walk (4 pairs per node visit), hot rows cached in LDS and averaged, "
                                  "cold rows device-scope atomics", "parallelism": "dp1"},
           "roofline": {"bound": "hbm", "achieved": round(wps * fabric_b / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(wps * fabric_b / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "w2v_hogwild_nm_kernel"}}
    rl = with_traffic(out["roofline"], "item2vec", "train", "w2v_hogwild_nm_kernel*", None, dt / steps * 1e3)
    rl["algorithmic_bytes"] = int(fabric_b * n)
    rl["model"] = {"bytes_per_word_fabric": round(fabric_b, 1), "bytes_per_word_all_rows": round(all_b, 1),
                   "mean_path_nodes": round(L_mean, 2), "mean_cold_path_nodes": round(Lc_mean, 2),
                   "cold_context_share": round(p_cold_ctx, 4), "chunks_per_position": chunks, "contexts_per_position": ctxs,
                   "hot_rows_per_table": hot_rows,
                   "basis": "the implemented node-major walk on this corpus' own Huffman paths and counts: cold rows = device-scope "
                            "load + atomic add of dim x 8 B each at the fabric; hot rows live in LDS between merges"}
    if rl.get("traffic"):
        rl["memory_side_GBs"] = rl.pop("hbm_side_GBs")
        rl["memory_side_frac"] = round(rl["memory_side_GBs"] / HBM_PEAK_GBS, 4)
        rl["memory_side_bytes_per_word"] = round(rl["traffic"] / n, 1)
    rl["l2"] = {"achieved": round(wps * all_b / 1e9, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(wps * all_b / 1e9 / L2_PEAK_GBS, 4),
                "hit_rate": rl.get("l2_hit_rate"), "note": "every row of the walk (LDS-resident ones included) against the aggregate L2 roof: an upper bound on what the L2s see"}
    sq = (pmc_entry("item2vec", "train", "w2v_hogwild_nm_kernel*") or {}).get("sq") or {}
    if sq.get("sq_insts_valu") and sq.get("kernel_cycles"):
        # a wave64 VALU instruction occupies its SIMD16 for 4 cycles; 256 CUs x 4 SIMDs
        rl["valu"] = {"issue_frac": round(sq["sq_insts_valu"] * 4.0 / (sq["kernel_cycles"] * 1024.0), 3),
                      "wave_instructions_per_word": round(sq["sq_insts_valu"] / n, 1),
                      "wait_any_pct_of_wave_cycles": sq.get("wait_any_pct_of_wave_cycles")}
    rl["binds"] = ("neither roof: the fabric sees < 0.1 of the HBM rate and the L2s < 0.1 of theirs.  The pass is LATENCY-bound: a stream's "
                   "walk is a chain of device-scope round trips (cold node: load -> inner product -> atomic add; 8 nodes prefetched), "
                   "the wavefronts wait 60+ % of their cycles (valu.wait_any_pct_of_wave_cycles) with 16 wavefronts per CU resident "
                   "(128 VGPRs each) -- more streams in flight, not more bandwidth, is what the kernel lacks")
    if not args.no_cpu_baseline:
        from oracle import pyoracle
        cores = usable_cores()
        pyoracle.set_threads(cores)
        cfg = pyoracle.w2v_cfg(dim=dim, optimizer="hs")
        paths = pyoracle.huffman_paths(counts)
        param = (np.random.default_rng(1).random((V, dim)) - 0.5) / dim
        aux = np.zeros((V - 1, dim))
        t0 = time.perf_counter()
        ns = 2_000_000
        pyoracle.w2v_train_hogwild(cfg, doc[:ns], cores, None, param, aux, paths, pyoracle.sigmoid_table(), 0.025, ns)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(ns / dtc, 1), "unit": "words/s", "cores": cores, "kind": "port",
                               "sample": f"one Hogwild pass over the first 2x10^6 words of the same corpus, oracle/orc_w2v.c with "
                                         f"{cores} threads, {dtc:.1f} s"}
    _emit(out)


def bench_knn(args):
    """Searcher.Search (search.go:92-134), brute-force cosine top-10 over V = 10^6 item vectors of
    D = 16 float64 (128 MB resident), 64 queries per call; a step = one call (scores + selection + D2H of the results)."""
    from goctr_amd import capi, search as gs
    capi.init(0)
    V, D, Q, k = 1_000_000, 16, 64, 10
    rng = np.random.default_rng(42)
    items = rng.standard_normal((V, D))
    s = gs.Searcher([""] * V, items)
    queries = rng.standard_normal((Q, D))
    # (a step = one call; the median of --regions back-to-back regions of --steps calls each: one region of 20 calls, as until round 5,
    # is a 1 ms sample -- two of eleven such runs read 1.17 M instead of 1.27 M queries/s, profiles/r05_knn_scan_threshold.txt)
    steps, warm = max(1, args.steps), max(1, args.warmup)
    for _ in range(warm):
        s.search_vectors(queries, k)
    capi.sync()
    # (round 6: the "two slow regions" of round 5's lines -- regions 1 and 5 of nine in EVERY process, +1.5 ms each -- are the
    # HARNESS: a call allocates a few dozen Python containers, and CPython's oldest-generation collection falls due every ~800
    # calls and takes ~1.5 ms over this process' heap (profiles/r06_knn_slow_regions.txt: GOCTR_BENCH_GC=1 brings them back).
    # A Go host has no such pause per 800 calls; the timed regions run with the collector paused.)
    import gc
    gc_off = os.environ.get("GOCTR_BENCH_GC", "0") != "1"
    if gc_off:
        gc.collect()
        gc.disable()
    regions = []
    for r in range(max(args.regions, 1)):
        t0 = time.perf_counter()
        for _ in range(steps):
            idx, sim, cnt = s.search_vectors(queries, k)
        capi.sync()
        regions.append(time.perf_counter() - t0)
    if gc_off:
        gc.enable()
    dt_py = sorted(regions)[(len(regions) - 1) // 2]
    regions_py = regions
    # The same closed loop from a COMPILED host (goctr_amd/host/knn_bench.cpp: C++ above the C-ABI, the catalogue and queries drawn the
    # same way): this Python loop pays ~10 us of interpreter per call (numpy allocations, ctypes marshalling) on top of a ~40 us call;
    # the reference's host is Go.  `value` is the compiled host's figure when the binary is there, the Python loop's is on the line
    # beside it (`python_loop`).
    compiled = None
    exe = os.path.join(ROOT, "goctr_amd", "host", "knn_bench")
    # (not under the profiler: scripts/prof_workload.sh passes --no-serving, and a child process of a rocprofv3 --pmc run competes with
    # its parent for the counters -- session r06_prof1 lost 40 minutes to ten 300-second time-outs)
    if os.path.exists(exe) and os.environ.get("GOCTR_BENCH_KNN_COMPILED", "1") != "0" and not args.no_serving:
        import subprocess
        try:
            capi.sync()
            r = subprocess.run([exe, "--items", str(V), "--dim", str(D), "--queries", str(Q), "--k", str(k), "--steps", str(steps),
                                "--warmup", str(max(warm, 200)), "--regions", str(max(args.regions, 1))], capture_output=True, text=True, timeout=300)
            compiled = json.loads(r.stdout.strip().splitlines()[-1])
            # (64 queries per call is this line's choice; the same loop at 256 per call beside it)
            r2 = subprocess.run([exe, "--items", str(V), "--dim", str(D), "--queries", "256", "--k", str(k), "--steps", str(steps),
                                 "--warmup", "200", "--regions", "5"], capture_output=True, text=True, timeout=300)
            compiled["at_256_queries_per_call"] = json.loads(r2.stdout.strip().splitlines()[-1])
        except Exception as e:                       # noqa: BLE001
            compiled = None
            print(f"bench.py: knn_bench failed ({e}); reporting the Python loop", file=sys.stderr)
    if compiled:
        regions = [x * 1e-3 for x in compiled["timed_regions_ms"]]
    dt = sorted(regions)[(len(regions) - 1) // 2]
    qps = steps * Q / dt
    # ---- roofline of the IMPLEMENTED call (rounds 1-4 priced the reference's loop -- every query scans
    # V x D x 8 bytes -- which the filter + refine path does not perform: the fraction read 12).  Per call of Q queries:
    #   scan     every normalised row ONCE, as two bf16 planes: V x D x 4 B; written: the maxima of every 32-item sub-block and of
    #            every 1024-item tile, per query: (V / 32 + V / 1024) x Q x 4 B
    #   collect  per query (one workgroup): the tile maxima, the sub-block maxima of the listed tiles, the float32 rows of the
    #            listed SUB-BLOCKS (the k-th largest of 256 group maxima sits near the (k + 3)-th best item: ~k + 3 sub-blocks of
    #            32 rows), the float64 rows of the few survivors; candidates and the replay stay in LDS
    listed = k + 3
    scan_b = V * D * 4 + (V // 32 + V // 1024 + 2) * Q * 4
    call_b = scan_b + Q * ((V // 1024 + 1) * 4 + listed * 32 * 4 + listed * 32 * D * 4 + listed * (D * 8 + 8))
    call_us = dt / steps * 1e6
    out = {"metric": "k-NN search queries/sec (cosine top-10 over 10^6 x 16 float64 items)", "value": round(qps, 1),
           "unit": "queries/s", "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": round(dt / steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "timed_regions": len(regions), "timed_regions_ms": [round(x * 1e3, 4) for x in regions],
           "harness": ("goctr_amd/host/knn_bench (C++ above the C-ABI: a closed loop of goctr_searcher_search calls, no interpreter between them)"
                       if compiled else "this Python loop (ctypes)"),
           "at_256_queries_per_call": ({k2: compiled["at_256_queries_per_call"][k2] for k2 in ("us_per_call", "queries_per_s", "timed_regions_ms")}
                                       if compiled and compiled.get("at_256_queries_per_call") else None),
           "python_loop": {"value": round(steps * Q / dt_py, 1), "ms_per_step": round(dt_py / steps * 1e3, 3),
                           "timed_regions_ms": [round(x * 1e3, 4) for x in regions_py],
                           "note": "the same calls driven from CPython: + the interpreter's per-call cost (numpy allocations, ctypes marshalling)"},
           "config": {"workload": "SURVEY 8(f)2: Searcher.Search, V=10^6, D=16 f64, k=10, 64 queries per call", "parallelism": "dp1"},
           "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                        "kernel": "knn_scan_bf16_kernel (the filter: every normalised row once per 64-query call, as two bf16 planes)"}}
    rl = with_traffic(out["roofline"], "knn", "train", "knn_scan_*", None, None)      # (knn_scan_bf16_kernel<D> from 12 queries per call on)
    rl["algorithmic_bytes"] = int(scan_b)                                              # the dominant kernel's own bytes
    if rl.get("avg_us_rocprofv3"):
        rl["achieved"] = round(rl["algorithmic_bytes"] / (rl["avg_us_rocprofv3"] * 1e-6) / 1e9, 1)
        rl["frac"] = round(rl["achieved"] / HBM_PEAK_GBS, 4)
        rl["duration_basis"] = "rocprofv3 average duration of the scan kernel (committed summary of this command)"
    rl["call"] = {"bytes_per_call_implemented": int(call_b), "us_per_call": round(call_us, 1),
                  "achieved_GBs": round(call_b / (call_us * 1e-6) / 1e9, 1), "frac_of_hbm": round(call_b / (call_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                  "note": "whole call (the queries stored by the host into device memory over the PCIe BAR -- a staged copy where there is "
                          "no large BAR --, scan, collect + replay, results through pinned memory, host wait) against the bytes the "
                          "implemented path has to move; the call is a latency chain of two dependent launches and a host round trip, "
                          "not a bandwidth problem: see the per-kernel durations in profiles/"}
    rl["reference_loop_bytes_per_call"] = int(Q * V * D * 8)   # what search.go:92-134 reads (every query scans the f64 matrix)
    if not args.no_cpu_baseline:
        from oracle import pyoracle
        norms = np.sqrt((items * items).sum(1))
        cores = usable_cores()
        pyoracle.set_threads(cores)
        pyoracle.knn_search_batch(items, queries[:cores], k, norms)                  # (warm-up: pages, threads)
        t0 = time.perf_counter()
        nq = 0
        qs = np.concatenate([queries] * 16)                                          # 1024 queries per batch, until ~12 s have passed
        while time.perf_counter() - t0 < 12.0:
            pyoracle.knn_search_batch(items, qs, k, norms)
            nq += qs.shape[0]
        dtc = time.perf_counter() - t0
        pyoracle.set_threads(1)
        out["cpu_baseline"] = {"value": round(nq / dtc, 2), "unit": "queries/s", "cores": cores, "kind": "port",
                               "sample": f"{nq} queries through oracle/orc_search.c (the reference's sequential loop per query, OpenMP over "
                                         f"the queries on {cores} threads = the container's CPU quota), {dtc:.1f} s"}
    _emit(out)


def bench_single_process(args):
    """`--gpus N --single-process`: ONE process, N engines (goctr_init_devices), every timed call a single goctr_train_steps with
    cfg.devices = N and the GLOBAL batch N x B -- rank r steps rows [r, r + 1) x B of every global batch on its replica, one
    all-reduce of the flat gradient per step, the identical Adam everywhere (csrc/ctr.hip train_multi).  Same line as the
    one-process-per-GPU run (weak scaling: B rows per GPU per step); the recommend / roofline / serving legs are the N = 1 run's."""
    from goctr_amd import capi, model as gm
    c = CFG
    N = args.gpus
    ids = [int(x) for x in os.environ.get("GOCTR_BENCH_DEVICES", ",".join(str(k) for k in range(N))).split(",")]
    if len(ids) != N:
        print(f"bench.py: GOCTR_BENCH_DEVICES names {len(ids)} devices, --gpus {N}", file=sys.stderr)
        sys.exit(2)
    capi.init_devices(ids)
    emb, ub, it, uf, cf, y = synth(args.rows * N, 42)
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, y)
    m = (gm.YoutubeDnn if c["KIND"] == "youtube" else gm.DinNet)(c["U"], c["T"], c["D"], c["D"], c["C"])
    init_weights(m, 1, 0.05 if args.train_emb > 0 else 1.0)
    pdrop = 0.003 if c["KIND"] == "youtube" else 0.005
    cfg = capi.default_train_cfg(batch=c["B"] * N, epochs=1, dropout_mode=2, p0=pdrop, p1=pdrop, seed=42, devices=N)
    if args.train_emb > 0:
        m.set_embedding_training(args.train_emb)
    gm.train_steps(m, ds, cfg, args.warmup, emb=tab)            # (replicas, shards, graphs: built here, outside the timed regions)
    capi.sync()
    regions, per_rank = [], []
    for r in range(max(args.regions, 1)):
        capi.sync()
        t0 = time.perf_counter()
        gm.train_steps(m, ds, cfg, args.steps, first_batch=args.warmup + r * args.steps, emb=tab)
        capi.sync()
        regions.append(time.perf_counter() - t0)
        # every rank's own span of the call on its own stream (goctr_engine_call_ms): the counterpart of the per-process wall times
        # of the one-process-per-GPU line
        try:
            per_rank.append([capi.engine_call_ms(k) / 1e3 for k in range(N)])
        except Exception:                       # noqa: BLE001
            per_rank.append([regions[-1]] * N)
    order = sorted(range(len(regions)), key=lambda i: regions[i])
    med = order[(len(order) - 1) // 2]
    dt = regions[med]
    # replicas must hold the same bits (same all-reduced gradient, same Adam)
    import zlib
    crcs = []
    for k in range(N):
        rep = m.replica(k) if k else m
        crcs.append(zlib.crc32(np.concatenate([rep.get_weights(nm).ravel() for nm in ("mlp0", "mlp1", "mlp2")]).tobytes()))
    out = {"metric": "training samples/sec (%s, MovieLens-20M-shaped synthetic)" % ("YouTube-DNN" if c["KIND"] == "youtube" else "DIN"),
           "value": round(args.steps * c["B"] * N / dt, 1), "unit": "samples/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": ("BASELINE configs[3]" if c["KIND"] == "youtube" else "BASELINE configs[2]") +
                                  f": batch {c['B']} per GPU, id mode, Dropout({pdrop}); SINGLE PROCESS, {N} engines on devices {ids}",
                      "global_batch": c["B"] * N, "parallelism": f"dp{N}", "resident_rows_per_gpu": args.rows},
           "dp_mode": "one process, goctr_init_devices + cfg.devices (RCCL ncclCommInitAll over distinct devices; loop-back communicator "
                      "when a device id repeats)",
           "timed_regions": len(regions), "timed_regions_ms": [round(x * 1e3, 4) for x in regions],
           "per_rank_ms_per_step": [round(x / args.steps * 1e3, 4) for x in per_rank[med]],
           "per_rank_basis": "device time of each rank's steps in the median region (events on the rank's own stream)",
           "replicas_bit_identical": len(set(crcs)) == 1}
    print(json.dumps(out), flush=True)
    if not out["replicas_bit_identical"]:
        print(f"bench.py: the {N} replicas DIVERGED (weight checksums {crcs})", file=sys.stderr)
        sys.exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=1 << 18, help="resident sample rows per GPU")
    ap.add_argument("--regions", type=int, default=9,
                    help="timed regions of exactly --steps steps, back to back; the line reports the median region (and lists all)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N > 1 WITHOUT one process per GPU: this process drives N engines (goctr_init_devices + cfg.devices = N, "
                         "the mode a single Go host uses: recommend.Train reaches N GPUs in one call, INTEGRATION.md 5.1).  "
                         "GOCTR_BENCH_DEVICES=0,0 names the device list (a repeated id = logical ranks on one GPU, loop-back communicator)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-serving", action="store_true",
                    help="skip the two boundary-inclusive recommend QPS figures (recommend_qps_keys: goctr_batch_predict from sample "
                         "keys; recommend_qps_host_rows: goctr_predict_dense from dense TrainSample rows in HOST memory)")
    ap.add_argument("--workload", default="din", choices=["din", "youtube", "mlp", "mlp100k", "item2vec", "knn"],
                    help="din = BASELINE configs[2] (the headline metric, default); youtube = configs[3] per-GPU slice "
                         "(10M x 64 table: the HBM-bound gather); mlp = configs[1]; mlp100k = configs[0] (the reference's own "
                         "MovieLens-100k run: 79 948 rows, batch 200, 20 epochs); item2vec = configs[4] per-GPU slice")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling (SURVEY 8(e) row 1): the GLOBAL batch stays BASELINE's (8192 DIN / 16384 YouTube) and every "
                         "rank steps batch / N rows of it; default: weak scaling, the per-GPU batch stays BASELINE's")
    ap.add_argument("--phase", default="all", choices=["all", "train", "predict"],
                    help="profiling passes (scripts/prof_workload.sh): run ONLY the training steps or ONLY the resident-row predict "
                         "batches, so that a rocprofv3 pass sees the launches of one phase (a kernel's training and predict launches "
                         "can have the same grid)")
    ap.add_argument("--train-emb", type=float, default=0.0, metavar="LR",
                    help="din / youtube: also train the embedding table (EXTENSION with no reference counterpart: SGD "
                         "scatter-add, csrc/emb_train.h).  Off by default: the headline metric keeps the reference's frozen table")
    args = ap.parse_args()
    if args.workload == "mlp":
        return bench_mlp(args)
    if args.workload == "mlp100k":
        return bench_mlp100k(args)
    if args.workload == "item2vec":
        return bench_item2vec(args)
    if args.workload == "knn":
        return bench_knn(args)
    if args.workload == "youtube":
        # BASELINE configs[3] / SURVEY 8(d) cfg4: YouTube-DNN, V = 10^7, D = 64, B = 16384 per GPU
        CFG.update(D=64, V=10_000_000, B=16384, KIND="youtube")

    if args.strong:
        if CFG["B"] % args.gpus:
            print(f"bench.py: --strong needs the global batch {CFG['B']} to be a multiple of --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)
        CFG["B_GLOBAL"] = CFG["B"]
        CFG["B"] //= args.gpus                    # every rank steps its share of BASELINE's global batch
    if args.single_process and args.gpus > 1 and "RANK" not in os.environ:
        return bench_single_process(args)
    from goctr_amd import launch
    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher around us: become one.  Rank 0 prints the JSON line.
        sys.exit(launch.spawn_local(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                    timeout=float(os.environ.get("GOCTR_BENCH_TIMEOUT", "1800"))))
    rank, world, local_rank = launch.env_rank()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks", file=sys.stderr)
        sys.exit(2)

    from goctr_amd import capi, model as gm
    capi.init(local_rank)             # (fails loudly here, before any rank waits on another, when there is no GPU)
    L = capi.load()
    rdv = launch.Rendezvous(rank, world)
    rccl_world = launch.init_comm(rdv, local_rank)
    if rccl_world != args.gpus:
        # a run that claims N GPUs but exchanges gradients among fewer is not the run that was asked for
        print(f"bench.py: --gpus {args.gpus} but the RCCL communicator has {rccl_world} ranks", file=sys.stderr)
        sys.exit(2)

    def barrier():
        capi.sync()
        rdv.barrier()

    def max_over_ranks(x: float) -> float:
        return rdv.max(x)

    c = CFG
    emb, ub, it, uf, cf, y = synth(args.rows, 42 + rank)
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, y)
    m = (gm.YoutubeDnn if c["KIND"] == "youtube" else gm.DinNet)(c["U"], c["T"], c["D"], c["D"], c["C"])
    # same weights on every rank.  With --train-emb the weights are 0.05 N(0,1): the reference's N(0,1) init saturates the
    # sigmoids, most row gradients underflow to exactly 0 and the scatter-add (which skips zeros) would look cheaper than
    # it is on a model that is actually learning
    init_weights(m, 1, 0.05 if args.train_emb > 0 else 1.0)
    # the reference ALWAYS trains with Dropout (0.005 DIN, din.go:204-205,307-312; 0.003 YouTube, dnn.go:136-137): the timed
    # step applies it too (counter-hash masks; GOCTR_BENCH_DROPOUT=0 for A/B experiments only)
    pdrop = 0.003 if c["KIND"] == "youtube" else 0.005
    cfg = capi.default_train_cfg(batch=c["B"], epochs=1, dropout_mode=2, p0=pdrop, p1=pdrop, seed=42)
    if os.environ.get("GOCTR_BENCH_DROPOUT", "1") == "0":
        cfg.dropout_mode = 0
    if args.train_emb > 0:
        m.set_embedding_training(args.train_emb)

    # ---- recommend QPS: rows scored per second through the predict path (PredBatch 4096).  Measured FIRST, on every rank,
    # over at least 8000 batches (~140 ms): an MI355X that has idled for >= 10 ms (the set-up above) runs its next
    # milliseconds ~6 % slower (scripts/launch_latency.py: 20 training steps take 60.5 instead of 57.0 us each, and 5
    # warm-up steps do not change that), so the short timed regions below start on a GPU that is already under load --
    # the state a training run is in for all but its first milliseconds.  Round 5 (the nine regions make it visible): behind
    # 2000 batches (~35 ms) the regions of a --steps 20 run still fell from 1.02 to 0.97 ms, behind 8000 they are flat at
    # 0.94-0.95 ms (and 30 000 change nothing more): the clocks need ~100 ms of load (GOCTR_BENCH_PRED_BATCHES).
    if args.phase != "predict":
        gm.train_steps(m, ds, cfg, 0, emb=tab)      # zero steps: allocates the workspace and captures the step graphs (host work)
    pred_batches = max(args.steps, int(os.environ.get("GOCTR_BENCH_PRED_BATCHES", "8000"))) if args.phase == "all" else args.steps * 4
    qps = None
    if args.phase != "train":
        gm.predict_steps(m, ds, c["PRED_B"], min(args.warmup, 20), emb=tab)
        barrier()
        t0 = time.perf_counter()
        gm.predict_steps(m, ds, c["PRED_B"], pred_batches, emb=tab)
        barrier()
        dtp = max_over_ranks(time.perf_counter() - t0)
        qps = pred_batches * c["PRED_B"] * world / dtp
    if args.phase == "predict":
        if rank == 0:
            print(json.dumps({"phase": "predict", "recommend_qps": round(qps, 1), "recommend_batch": c["PRED_B"], "batches": pred_batches,
                              "workload": c["KIND"]}), flush=True)
        rdv.barrier()
        rdv.close()
        return

    # ---- training samples/sec: W warm-up steps, then exactly K timed steps -- R times back to back (same K, same graphs, each
    # region bracketed by sync + barrier on both sides and reduced with MAX over the ranks), reported from the MEDIAN region.
    # One region at the driver's flags is a single ~1 ms window; a fresh lease's first millisecond after the predict leg has
    # moved the headline by 5 % between rounds with byte-identical kernels.  Every region is on the line.
    # ---- pre-load in the timed load's own kernels.  The predict leg above is a gather-bound load; behind it the first ~9 ms of
    # TRAINING steps are still up to 3.5 % slower than the rest (profiles/r05_bench_preload.txt: nine 20-step regions fell from
    # 0.985 to 0.951 ms on one box, from 1.003 to 0.974 ms on another) -- the clocks follow the kind of load, not only its
    # presence.  2000 training steps (~95 ms) of a SCRATCH model of the same shape on the same data come first; then the W
    # warm-up steps of the model that is measured, then the regions, which are flat behind it.  GOCTR_BENCH_PRELOAD_STEPS=0: off.
    # (One GPU only: a second model stepping through the RCCL communicator is a path the multi-GPU runs have never taken, and a
    # scaling run is not the place to take it first; there the step also waits on the all-reduce, not only on the clocks.)
    # Symmetry across N: the pre-load only exists at N = 1, so an N = 1 line is clock-warmed by 2000 training
    # steps and an N > 1 line is not.  The N = 1 run therefore measures BOTH: first the W warm-up steps and the nine regions
    # exactly as an N > 1 run takes them (`without_preload` on the line: the figure a scaling curve must be read against),
    # then the pre-load and nine more regions (the headline).
    preload = int(os.environ.get("GOCTR_BENCH_PRELOAD_STEPS", "2000")) if args.phase == "all" and world == 1 else 0
    cursor = [0]

    def timed_regions():
        regs, per_rank = [], []
        for _ in range(max(args.regions, 1)):
            barrier()
            t0 = time.perf_counter()
            gm.train_steps(m, ds, cfg, args.steps, first_batch=cursor[0], emb=tab)
            barrier()
            dt_local = time.perf_counter() - t0
            cursor[0] += args.steps
            per_rank.append(rdv.allgather(dt_local))
            regs.append(max_over_ranks(dt_local))
        return regs, per_rank

    def median_index(regs):
        order = sorted(range(len(regs)), key=lambda i: regs[i])
        return order[(len(order) - 1) // 2]          # (lower median for an even count: never an average of two regions)

    gm.train_steps(m, ds, cfg, args.warmup, emb=tab)
    cursor[0] = args.warmup
    without_preload = None
    m_pre = None
    if preload > 0:
        regs0, _ = timed_regions()
        d0 = regs0[median_index(regs0)]
        without_preload = {"value": round(args.steps * c["B"] * world / d0, 1), "ms_per_step": round(d0 / args.steps * 1e3, 4),
                           "timed_regions_ms": [round(x * 1e3, 4) for x in regs0],
                           "note": "the same W warm-up steps and regions WITHOUT the scratch-model pre-load in front: what an N > 1 run "
                                   "(which never pre-loads) is comparable with"}
        m_pre = (gm.YoutubeDnn if c["KIND"] == "youtube" else gm.DinNet)(c["U"], c["T"], c["D"], c["D"], c["C"])
        init_weights(m_pre, 2, 1.0)
        gm.train_steps(m_pre, ds, cfg, preload, emb=tab)      # (never trains the embedding table: --train-emb is set on `m` only)
        barrier()
        gm.train_steps(m, ds, cfg, args.warmup, first_batch=cursor[0], emb=tab)
        cursor[0] += args.warmup
    regions, per_rank_regions = timed_regions()
    med = median_index(regions)
    dt = regions[med]
    per_rank_ms = [round(x / args.steps * 1e3, 4) for x in per_rank_regions[med]]
    samples_per_s = args.steps * c["B"] * world / dt

    out = {
        "metric": "training samples/sec (%s, MovieLens-20M-shaped synthetic)" % ("YouTube-DNN" if c["KIND"] == "youtube" else "DIN"),
        "value": round(samples_per_s, 1),
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[3] per-GPU slice: YouTube-DNN (mean pooling), T=50, D=64, U=52, C=53, vocab 10^7 "
                                "(2.56 GB table replicated per GPU, frozen = reference semantics), batch 16384 per GPU, id mode"
                                if c["KIND"] == "youtube" else
                                "BASELINE configs[2]: DIN cosine attention, T=50, D=16, U=52, C=53, vocab 26744, "
                                "batch 8192 per GPU, id mode (keys + table resident in HBM)") +
                               (f"; Dropout({pdrop}) on both hidden layers like the reference" if cfg.dropout_mode else "; dropout OFF (experiment)") +
                               (f"; STRONG scaling: global batch {c['B'] * world} split over {world} ranks, {c['B']} rows per GPU per step" if args.strong else ""),
                   "numerics": ("float32 in, float32 out; every GEMM of the training step (layer chain and weight gradients) on the "
                                "6-product bf16 split with float32 accumulation -- each float32 operand is the exact sum of three bf16 "
                                "planes, a*b = hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid); measured MORE accurate than the "
                                "f32 MFMA (2.8e-8 vs 1.3e-7 of sum|ab|, scripts/ubench/bf16x3.hip) and bounded in tests/ by the float32 "
                                "oracle's own error against a float64 evaluation; predict launches of >= 8192 rows run the forward-only variant "
                                "of the same bf16-split chain, smaller ones ctr_fwd16_kernel on v_mfma_f32_16x16x4_f32 "
                                "(GOCTR_CHAIN_X3=0 / GOCTR_TN_F32=1 / GOCTR_PREDICT_X3=0 select the f32 MFMA bodies)"),
                   "global_batch": c["B"] * world, "parallelism": f"dp{world}", "resident_rows_per_gpu": args.rows},
        "recommend_qps": round(qps, 1) if qps else None, "recommend_batch": c["PRED_B"],
        "recommend_note": f"PredBatchSize {c['PRED_B']} at the API; the engine scores {os.environ.get('GOCTR_PRED_GROUP', '8')} consecutive batches per launch "
                          "(at most 32768 rows), on the forward-only bf16-split chain, one persistent workgroup per CU over the launch's row tiles (rows are scored "
                          "independently; scores equal one-batch launches to float32 rounding; GOCTR_PRED_GROUP=1 for one batch per launch)",
        "rccl_world": rccl_world, "per_rank_ms_per_step": per_rank_ms,
        "timed_regions": len(regions), "timed_regions_ms": [round(x * 1e3, 4) for x in regions],
        "timed_region_basis": f"value / ms_per_step are the MEDIAN of {len(regions)} back-to-back regions of exactly {args.steps} steps each "
                              "(barrier + device sync on both sides of every region, max over ranks per region)",
        "timed_region_min_ms": round(min(regions) * 1e3, 4), "timed_region_max_ms": round(max(regions) * 1e3, 4),
        "timed_region_spread": round((max(regions) - min(regions)) / dt, 4),
        "preload": {"predict_batches": pred_batches, "scratch_model_training_steps": preload,
                    "note": "untimed device pre-load in front of the W warm-up steps: the recommend-QPS leg, then (N = 1 only) training "
                            "steps of a scratch model of the same shape; N = 1 lines carry `without_preload` = the same measurement "
                            "taken before that pre-load, the protocol every N > 1 line follows"},
        "without_preload": without_preload,
    }
    if world > 1:
        import ctypes as C
        mode = C.c_int(0)
        capi.check(L.goctr_comm_capture_mode(C.byref(mode)))
        out["dp_allreduce"] = ("a node of the multi-step graphs (captured RCCL collective, self-tested on this communicator at start-up; "
                               "before this run the captured form had only ever executed with world = 1: the builder's boxes have one GPU)"
                               if mode.value == 1 else "between graph launches (capture off or failed its self-test)")
        out["dp_mode"] = "one process per GPU (goctr_comm_init over the launcher's rendezvous); the single-process entry is goctr_init_devices + cfg.devices"
    if args.train_emb > 0:
        out["sparse_exchange_bytes_per_step_per_rank"] = m.sparse_exchange_bytes()      # (0 without a communicator)
        # the one-time sparse plan of the dataset (csrc/emb_plan.hip) is built before the timed region: charged here to a run of
        # E epochs over the resident rows -- samples/s = E rows / (E batches x step + plan build)
        pms, pnb = m.emb_plan_build_ms()
        step_s = dt / args.steps
        out["plan_build_ms"] = round(pms, 3)
        out["plan_build_us_per_batch"] = round(pms * 1e3 / max(pnb, 1), 2)
        out["samples_per_s_incl_plan"] = {f"epochs={E}": round(E * pnb * c["B"] * world / max(E * pnb * step_s + pms * 1e-3, 1e-12), 1) for E in (1, 20, 200)}
    if args.train_emb > 0:
        out["config"]["workload"] += (f"; EXTENSION: embedding table trained too (SGD scatter-add, lr {args.train_emb}; "
                                      "the reference keeps it frozen; weights 0.05 N(0,1) so that the row gradients are non-zero)")
        out["config"]["train_embeddings"] = True

    if rank == 0 and not args.no_serving:
        out.update(serving_qps(m, tab, emb, ub, it, uf, cf))
        if world == 1:
            capi.sync()
            out.update(rank_serving(c["KIND"]))      # (its own process: the GPU is idle here)

    # ---- roofline: instrumented re-run of the same K steps (eager, hipEvent pair per launch)
    if not args.no_roofline:
        # every rank repeats the K steps (the all-reduce needs all of them); rank 0 runs them eagerly
        # with an event pair around every launch
        if rank == 0:
            capi.prof_enable(True)
            capi.prof_reset()
        gm.train_steps(m, ds, cfg, args.steps, first_batch=args.warmup, emb=tab)
        capi.sync()
        if rank == 0:
            prof = capi.prof_get()
            capi.prof_enable(False)
            work = kernel_work(args.train_emb > 0)
            table = {k: {"avg_us": round(ms / n * 1e3, 2), "launches": n} for k, (ms, n) in prof.items() if n}
            wl = "youtube" if c["KIND"] == "youtube" else "din"
            if args.train_emb > 0:
                wl += "emb"                      # profiles/r02_{dinemb,youtubeemb}_*: the same command with --train-emb
                # algorithmic bytes of the id-major sparse row update per launch (emb_slot_kernel, batch 0's counts): every
                # valid (sample, slot) pair reads its 12-byte plan entry and the sample's dp row (DIN: + the coefficient, the
                # candidate row v and the slot's row x), every distinct id's row is read and written once; priced like the
                # gather, on memory-side bytes
                ids0 = np.concatenate([ub[:c["B"]].ravel(), it[:c["B"]]])
                ids0 = ids0[(ids0 >= 0) & (ids0 < c["V"])]
                pairs0, distinct0 = int(ids0.size), int(np.unique(ids0).size)
                per_pair = 12 + 4 * c["D"] + ((16 + 8 * c["D"]) if c["KIND"] == "din" else 0)
                work["emb_grad"] = ("hbm", pairs0 * per_pair + distinct0 * 8 * c["D"])
                out["emb_batch0"] = {"valid_pairs": pairs0, "distinct_ids": distinct0}
            syms = capi.prof_kernels()           # the kernel symbol each family's launches actually ran (goctr_prof_kernel)
            dom = max((k for k in table if k in work), key=lambda k: prof[k][0])
            kind, w = work[dom]
            dom_ms = prof[dom][0] / prof[dom][1]
            rl = roofline_obj(kind, w, dom_ms)
            rl["kernel"] = dom
            # launch shape of the timed launch (threads), where this harness knows it: the entry of the committed summary
            # must be the same symbol AND the same grid
            grid = {"chain": -(-c["B"] // 32) * 512, "attn_fwd": -(-c["B"] // 4) * 256}.get(dom)
            alg = chain_algorithmic_bytes(args.train_emb > 0) if dom == "chain" else (w if kind == "hbm" else None)
            rl = with_traffic(rl, wl, "train", syms.get(dom), grid, dom_ms, alg)
            if dom == "emb_grad":
                rl["algorithmic_GBs"] = rl["achieved"]
                if rl.get("hbm_side_GBs"):
                    rl["achieved"] = rl.pop("hbm_side_GBs")
                    rl["frac"] = round(rl["achieved"] / HBM_PEAK_GBS, 4)
                rl["note"] = "sparse scatter-add of the embedding-row gradients (DESIGN 4.10); priced on memory-side bytes"
            rl["algorithmic_bytes"] = chain_algorithmic_bytes(args.train_emb > 0) if dom == "chain" else None
            if dom == "chain" and attn_bwd_in_chain(args.train_emb > 0):
                rl["note"] = ("this launch also does the attention backward (round 2: a launch of its own, 6.0 us): 8192 x 50 row gathers and "
                              "dot products at its tail add ~3.8 us to it and no MFMA work, so its fraction of the MFMA peak reads lower "
                              "than the same products alone (0.37-0.39) while the step got 2.3 us shorter")
            rl["duration_basis"] = ("hipEvent pair around every launch of an eager re-run of the K steps (includes the launch gap: "
                                    "reads ~2-3 us above rocprofv3's kernel duration)")
            if rl.get("avg_us_rocprofv3") and kind == "mfma":      # the same work over rocprofv3's own average duration of that kernel (committed summary)
                rl["frac_at_rocprofv3_duration"] = round(w / (rl["avg_us_rocprofv3"] * 1e-6) / 1e12 / FP32_MFMA_PEAK_TF, 4)
            out["roofline"] = rl
            if "attn_fwd" in table:
                # the gather against the HBM roof, MEMORY-SIDE: bytes the memory system served per launch (PMC, the training
                # phase's eager attn_fwd launch of B rows) over the live duration of THAT SAME launch shape in this run.  The
                # algorithmic bytes (every row and id of every sample) are reported next to it; at cfg3 the 1.7 MB table is
                # L2-resident and the kernel is VALU-bound, at cfg4 (2.56 GB table) the Zipf-hot rows still hit in L2, so the
                # algorithmic rate overstates what HBM delivers.
                gk, gw = work["attn_fwd"]
                gms = prof["attn_fwd"][0] / prof["attn_fwd"][1]
                grl = roofline_obj(gk, gw, gms)
                grl["kernel"] = "attn_fwd (embedding gather + attention pooling), the training step's launch of B rows"
                grl["algorithmic_GBs"] = grl["achieved"]
                grl = with_traffic(grl, wl, "train", syms.get("attn_fwd"), -(-c["B"] // 4) * 256, gms, None)
                if grl.get("hbm_side_GBs"):
                    grl["achieved"] = grl.pop("hbm_side_GBs")
                    grl["frac"] = round(grl["achieved"] / HBM_PEAK_GBS, 4)
                    grl["basis"] = "memory-side bytes (PMC FETCH_SIZE x 2 + WRITE_SIZE) of this kernel symbol at this grid / live hipEvent duration of the same launch"
                else:
                    grl["basis"] = "algorithmic bytes (no rocprofv3 summary committed for this kernel symbol and grid)"
                out["gather_roofline"] = grl
            # the whole replayed step against its algorithmic bytes (committed summary: sum over the step's launches)
            try:
                import glob
                pf = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{wl}_kernels.json")))
                stt = json.load(open(pf[-1])).get("step_total") if pf else None
                if stt and stt.get("traffic_ratio") and stt.get("batch") == c["B"]:
                    out["step_traffic_ratio"] = stt["traffic_ratio"]
                    out["step_traffic"] = {"memory_side_bytes": stt["sum_hbm_bytes"], "algorithmic_bytes": stt["algorithmic_bytes"],
                                           "sum_kernel_us_rocprofv3": stt["sum_avg_us"], "kernels": stt["kernels"], "source": os.path.basename(pf[-1])}
            except Exception:
                pass
            out["kernel_symbols"] = {k: v for k, v in syms.items() if v and k in table}
            out["kernels"] = table
    rdv.barrier()

    if world > 1:
        # Data-parallel replicas must stay BIT-identical (same all-reduced gradient, same Adam; with --train-emb the same
        # integer row sums from the owners): every rank checksums its weights (and the first 65 536 rows of its table -- the
        # Zipf-hot ids are the low ones) and rank 0 compares.  This is the only place a W > 1 run of the device code is ever
        # checked: the GPU test boxes have one GPU.
        import ctypes as C
        import zlib
        capi.sync()
        crc = 0
        for tid, shape in ((0, (c["U"] + 2 * c["D"] + c["C"], c["H1"])), (1, (c["H1"], c["H2"])), (2, (c["H2"], 1))) + \
                (((3, (1, c["T"])),) if c["KIND"] == "din" else ()):
            w = np.zeros(shape, np.float32)
            capi.check(L.goctr_model_get_weights(m._h, C.c_int(tid), capi.ptr(w, C.c_float), C.c_size_t(w.size)))
            crc = zlib.crc32(w.tobytes(), crc)
        if args.train_emb > 0:
            nrow = min(c["V"], 65536)
            rows = np.zeros((nrow, c["D"]), np.float32)
            capi.check(L.goctr_emb_get_rows(tab._h, C.c_int64(0), C.c_int64(nrow), capi.ptr(rows, C.c_float)))
            crc = zlib.crc32(rows.tobytes(), crc)
        crcs = rdv.allgather(int(crc))
        out["replicas_bit_identical"] = len(set(crcs)) == 1
        if rank == 0 and not out["replicas_bit_identical"]:
            print(f"bench.py: the {world} data-parallel replicas DIVERGED (weight / table checksums {crcs})", file=sys.stderr)

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    rdv.barrier()
    if world > 1:
        L.goctr_comm_destroy()
    rdv.close()
    if world > 1 and not out.get("replicas_bit_identical", True):
        sys.exit(3)


if __name__ == "__main__":
    main()
