#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs written by scripts/prof_round.sh (gpurun_out/p/) into the summaries
committed under profiles/:  <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim),
<tag>_kernel_trace_by_grid.txt (training vs predict launches separated by grid size) and
<tag>_pmc_traffic.json (HBM bytes per launch from the separate --pmc FETCH_SIZE / WRITE_SIZE
passes; FETCH_SIZE doubled as MI355X_MICROARCH.md "HBM" prescribes for gfx950, both reported in KiB
by rocprofv3).  bench.py reads the JSON to fill roofline.traffic.

usage: python scripts/prof_summarize.py r01_v2 [gpurun_out/p]
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

SHORT = {"reduce_adam_kernel": "reduce", "attn_fwd_kernel": "attn_fwd", "ctr_chain_kernel": "chain", "attn_bwd_kernel": "attn_bwd",
         "gemm_tn_multi_kernel": "dW0", "reduce_kernel": "reduce", "adam_kernel": "adam",
         "gemm_nn_kernel": "gemm_nn", "gemm_tn_kernel": "gemm_tn"}


def short(name):
    for k, v in SHORT.items():
        if k in name:
            return v
    return None


def main():
    tag = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/p"
    os.makedirs("profiles", exist_ok=True)
    stats = glob.glob(f"{src}/kt/*/*_kernel_stats.csv")[0]
    shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
    if os.path.exists(f"{src}/kt_bench.json"):
        shutil.copy(f"{src}/kt_bench.json", f"profiles/{tag}_bench_under_rocprof.json")
    trace = glob.glob(f"{src}/kt/*/*_kernel_trace.csv")[0]
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        s = short(r["Kernel_Name"])
        if s:
            by[(s, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    grids = {}
    with open(f"profiles/{tag}_kernel_trace_by_grid.txt", "w") as f:
        f.write("# rocprofv3 --kernel-trace: per kernel and launch shape (threads, workgroup)\n")
        f.write(f"{'kernel':10s} {'grid':>8s} {'wg':>5s} {'calls':>6s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s}\n")
        for (s, g, w), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            f.write(f"{s:10s} {g:8d} {w:5d} {len(v):6d} {sum(v)/len(v)/1e3:8.2f} {min(v)/1e3:8.2f} {max(v)/1e3:8.2f}\n")
            grids.setdefault(s, []).append((g, sum(v) / len(v) / 1e3))
    traffic = {}
    for leg, key, corr in (("fetch", "fetch_bytes", 2.0), ("write", "write_bytes", 1.0)):
        files = glob.glob(f"{src}/{leg}/*/*_counter_collection.csv")
        if not files:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            s = short(r["Kernel_Name"])
            if s:
                acc[(s, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        for (s, g), v in acc.items():
            # the training launch of a kernel is its largest grid (predict batches are smaller)
            train_grid = max(gg for (ss, gg) in acc if ss == s)
            if g != train_grid:
                continue
            traffic.setdefault(s, {})[key] = sum(v) / len(v) * 1024.0 * corr
            traffic[s][key + "_raw_kib"] = sum(v) / len(v)
    for s, t in traffic.items():
        t["hbm_bytes"] = t.get("fetch_bytes", 0.0) + t.get("write_bytes", 0.0)
        for g, us in grids.get(s, []):
            pass
    out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, GOCTR_NO_GRAPH=1), "
                     "bench.py cfg3 training launches; FETCH_SIZE x2 (gfx950 correction), KiB -> bytes",
           "per_launch": traffic}
    json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    print(open(f"profiles/{tag}_kernel_trace_by_grid.txt").read())
    print(json.dumps({k: round(v["hbm_bytes"] / 1e6, 2) for k, v in traffic.items()}))


if __name__ == "__main__":
    main()
