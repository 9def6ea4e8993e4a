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

SHORT = {"reduce_adam_kernel": "reduce", "attn_fwd_kernel": "attn_fwd", "ctr_chain_x3_kernel": "chain", "ctr_chain_kernel": "chain",
         "attn_bwd_kernel": "attn_bwd", "gemm_tn_multi_x3w_kernel": "dW0", "gemm_tn_multi_x3_kernel": "dW0", "gemm_tn_multi_kernel": "dW0", "ctr_fwd16_kernel": "fwd16",
         "reduce_attn_kernel": "reduce_attn", "reduce_adam_kernel": "reduce", "reduce_kernel": "reduce", "adam_kernel": "adam", "gemm_nn_kernel": "gemm_nn", "gemm_tn_kernel": "gemm_tn"}


def short(name):
    """bench.py's family name for the DIN / YouTube step kernels; any other kernel of the library keeps its own
    function name (goctr::mlp_fwd_kernel<...>(...) -> mlp_fwd_kernel)"""
    for k, v in SHORT.items():
        if k in name:
            return v
    import re
    m = re.search(r"(?:goctr::)?([A-Za-z_][A-Za-z0-9_]*)\s*(?:<|\()", name.replace("void ", ""))
    if m and ("goctr" in name or m.group(1).endswith("_kernel")):
        return m.group(1)
    return None


def main():
    tag = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/p"
    os.makedirs("profiles", exist_ok=True)
    # gpurun merges a call's files INTO gpurun_out/: an earlier run's CSVs (other pid prefix) may still lie beside the new ones
    newest = lambda files: max(files, key=os.path.getmtime)
    stats = newest(glob.glob(f"{src}/kt/*/*_kernel_stats.csv") + glob.glob(f"{src}/kt/**/*_kernel_stats.csv", recursive=True))
    shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
    if os.path.exists(f"{src}/kt_bench.json"):
        shutil.copy(f"{src}/kt_bench.json", f"profiles/{tag}_bench_under_rocprof.json")
    trace = newest(glob.glob(f"{src}/kt/*/*_kernel_trace.csv") + glob.glob(f"{src}/kt/**/*_kernel_trace.csv", recursive=True))
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
        for r in csv.DictReader(open(newest(files))):
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
    # SQ pass: MFMA busy / issue counters per training launch (MfmaUtil = MFMA_BUSY / (GUI_ACTIVE * SIMDs))
    sqf = glob.glob(f"{src}/sq/*/*_counter_collection.csv")
    sq = {}
    if sqf:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(newest(sqf))):
            sname = short(r["Kernel_Name"])
            if sname:
                acc[(sname, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for (sname, g), v in acc.items():
            if g != max(gg for (ss, gg) in acc if ss == sname):
                continue
            m = {c: sum(x) / len(x) for c, x in v.items()}
            ent = {c.lower(): round(val) for c, val in m.items()}
            # SQ_BUSY_CYCLES is reported per shader engine and summed over the 32 SEs (8 XCDs x 4): /32 = the launch's
            # duration in shader cycles (cross-checks against the kernel-trace duration at ~2.0 GHz);
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs
            if m.get("SQ_BUSY_CYCLES"):
                ent["kernel_cycles"] = round(m["SQ_BUSY_CYCLES"] / 32)
                ent["mfma_busy_pct"] = round(100.0 * (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024) / (m["SQ_BUSY_CYCLES"] / 32), 1)
            ent["mfma_flops_f32"] = round(m.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512)
            sq[sname] = ent
        json.dump({"source": "rocprofv3 --pmc SQ_* (own pass, GOCTR_NO_GRAPH=1); mfma_busy_pct = (SQ_VALU_MFMA_BUSY_CYCLES / "
                             "1024 SIMDs) / (SQ_BUSY_CYCLES / 32 SEs)", "per_launch": sq},
                  open(f"profiles/{tag}_sq_counters.json", "w"), indent=1)
        print(json.dumps({k: (v.get("mfma_busy_pct"), v.get("mfma_flops_f32"), v.get("kernel_cycles")) for k, v in sq.items()}))
    # L2 pass: hit rate per kernel (MI355X_MICROARCH.md "L2": TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum))
    l2f = glob.glob(f"{src}/l2/*/*_counter_collection.csv")
    if l2f:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(newest(l2f))):
            sname = short(r["Kernel_Name"])
            if sname:
                acc[(sname, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for (sname, g), v in acc.items():
            if g != max(gg for (ss, gg) in acc if ss == sname):
                continue
            hit, miss = (sum(v.get(c, [0.0])) / max(1, len(v.get(c, [0.0]))) for c in ("TCC_HIT_sum", "TCC_MISS_sum"))
            t = traffic.setdefault(sname, {})
            t["l2_hits"], t["l2_misses"] = hit, miss
            t["l2_hit_rate"] = round(hit / (hit + miss), 4) if hit + miss > 0 else None
    # durations next to the byte counts, so that a reader (bench.py) can turn them into GB/s without the trace file
    for sname, lst in grids.items():
        if sname in traffic:
            traffic[sname]["avg_us_rocprof"] = round(max(lst)[1], 3)     # the largest grid = the training launch
    import subprocess
    try:
        head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        head = None
    out = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, GOCTR_NO_GRAPH=1: eager steps), training "
                     "launches (largest grid of each kernel); FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md 'HBM'), KiB -> "
                     "bytes; l2_* from a --pmc TCC_HIT_sum TCC_MISS_sum pass",
           "commit": head, "per_launch": traffic}
    json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    print(open(f"profiles/{tag}_kernel_trace_by_grid.txt").read())
    print(json.dumps({k: round(v["hbm_bytes"] / 1e6, 2) for k, v in traffic.items()}))


if __name__ == "__main__":
    main()
