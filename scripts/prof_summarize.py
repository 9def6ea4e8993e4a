#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs written by scripts/prof_workload.sh (gpurun_out/p_<name>/) into the summaries committed under
profiles/:

  <tag>_kernel_stats.csv          rocprofv3 --kernel-trace --stats of the TRAINING phase, verbatim
  <tag>_predict_kernel_stats.csv  the same for the PREDICT phase (workloads that have one)
  <tag>_kernels.json              per (phase, kernel symbol incl. template arguments, grid): calls, avg / min / max duration,
                                  memory-side bytes (separate --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE x 2 as
                                  MI355X_MICROARCH.md "HBM" prescribes for gfx950; rocprofv3 reports KiB), L2 hit rate
                                  (TCC_HIT_sum / TCC_MISS_sum pass), SQ counters (MFMA-busy, wave cycles, wait cycles)
  <tag>_kernel_trace.txt          the same table, human-readable

Every entry is keyed on the PHASE the pass ran (bench.py --phase train / --phase predict: separate processes, so a
training launch can never be mistaken for a predict launch of the same grid) and on the kernel's FULL symbol --
ctr_chain_x3_kernel<9,false> and <9,true> are different entries.  bench.py looks an entry up by (phase, symbol) and
refuses it when the symbol is not the one it just timed.

usage: python scripts/prof_summarize.py r03_din gpurun_out/p_din
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys


def symbol(name):
    """'void goctr::ctr_chain_x3_kernel<9, false>(goctr::ChainX3Args)' -> 'ctr_chain_x3_kernel<9,false>' (None for kernels
    that are not this library's)"""
    n = name.strip().strip('"')
    if n.endswith(".kd"):
        n = n[:-3]
    n = re.sub(r"^void\s+", "", n)
    n = n.replace("(anonymous namespace)::", "")
    # cut the argument list: the last top-level '(' of the name
    depth, cut = 0, None
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    if cut is not None:
        n = n[:cut]
    if "goctr" not in name and not n.endswith("_kernel") and "_kernel<" not in n:
        return None
    n = n.replace("goctr::", "")
    return n.replace(" ", "")


def newest(files):
    return max(files, key=os.path.getmtime) if files else None


def find(src, leg, suffix):
    return newest(glob.glob(f"{src}/{leg}/**/*{suffix}", recursive=True))


def trace_table(path):
    by = collections.defaultdict(list)
    if not path:
        return by
    for r in csv.DictReader(open(path)):
        s = symbol(r["Kernel_Name"])
        if s:
            # (the counter CSVs report the TOTAL grid; the trace reports it per dimension)
            grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
            by[(s, grid, int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return by


def counter_table(path):
    """{(symbol, grid): {counter: mean over the launches}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if not path:
        return {}
    for r in csv.DictReader(open(path)):
        s = symbol(r["Kernel_Name"])
        if s:
            acc[(s, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def summarize_phase(src, phase, legs):
    """legs: {'kt': dir, 'fetch': dir, 'write': dir, 'sq': dir, 'l2': dir} (missing ones are skipped)"""
    out = {}
    tr = trace_table(find(src, legs["kt"], "_kernel_trace.csv")) if "kt" in legs else {}
    for (s, g, w), v in tr.items():
        out[f"{s}@{g}"] = {"kernel": s, "grid_threads": g, "workgroup": w, "calls": len(v), "avg_us": round(sum(v) / len(v) / 1e3, 3),
                           "min_us": round(min(v) / 1e3, 3), "max_us": round(max(v) / 1e3, 3)}
    fetch = counter_table(find(src, legs["fetch"], "_counter_collection.csv")) if "fetch" in legs else {}
    write = counter_table(find(src, legs["write"], "_counter_collection.csv")) if "write" in legs else {}
    l2 = counter_table(find(src, legs["l2"], "_counter_collection.csv")) if "l2" in legs else {}
    sq = counter_table(find(src, legs["sq"], "_counter_collection.csv")) if "sq" in legs else {}
    for tab in (fetch, write, l2, sq):
        for (s, g) in tab:
            out.setdefault(f"{s}@{g}", {"kernel": s, "grid_threads": g})
    for key, ent in out.items():
        k = (ent["kernel"], ent["grid_threads"])
        if k in fetch and "FETCH_SIZE" in fetch[k]:
            ent["fetch_raw_kib"] = round(fetch[k]["FETCH_SIZE"], 2)
            ent["fetch_bytes"] = round(fetch[k]["FETCH_SIZE"] * 1024.0 * 2.0)
        if k in write and "WRITE_SIZE" in write[k]:
            ent["write_raw_kib"] = round(write[k]["WRITE_SIZE"], 2)
            ent["write_bytes"] = round(write[k]["WRITE_SIZE"] * 1024.0)
        if "fetch_bytes" in ent and "write_bytes" in ent:
            ent["hbm_bytes"] = ent["fetch_bytes"] + ent["write_bytes"]
        if k in l2:
            hit, miss = l2[k].get("TCC_HIT_sum", 0.0), l2[k].get("TCC_MISS_sum", 0.0)
            ent["l2_hits"], ent["l2_misses"] = round(hit), round(miss)
            ent["l2_hit_rate"] = round(hit / (hit + miss), 4) if hit + miss > 0 else None
        if k in sq:
            m = sq[k]
            e = {c.lower(): round(v) for c, v in m.items()}
            # SQ_BUSY_CYCLES is reported per shader engine and summed over the 32 SEs (8 XCDs x 4): / 32 = the launch's
            # duration in shader cycles; SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs
            if m.get("SQ_BUSY_CYCLES"):
                e["kernel_cycles"] = round(m["SQ_BUSY_CYCLES"] / 32)
                e["mfma_busy_pct"] = round(100.0 * (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024) / (m["SQ_BUSY_CYCLES"] / 32), 1)
            if m.get("SQ_WAVE_CYCLES"):
                e["wait_any_pct_of_wave_cycles"] = round(100.0 * m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"], 1)
            ent["sq"] = e
    return out


# the launches of ONE graph-replayed training step per workload (what bench.py's `value` times), and the step's algorithmic bytes
# per sample (cfg3 whole step ~3 900 B; cfg4: rows 13 056 + ids 204 + side features 420 = 13 680 B)
STEP_KERNELS = {"din": (("ctr_chain_x3_kernel<", ",false>"), ("gemm_tn_multi_x3w_", ""), ("reduce_attn_kernel<", "")),
                "youtube": (("ctr_chain_x3_kernel<", ",false>"), ("gemm_tn_multi_x3w_", ""), ("reduce_attn_kernel<", ""))}
STEP_ALG_BYTES_PER_SAMPLE = {"din": 3900, "youtube": 13680}
STEP_BATCH = {"din": 8192, "youtube": 16384}


def step_total(tab, workload):
    """sum over the replayed step's launches: rocprofv3 duration and memory-side bytes, each kernel at its most-called shape"""
    if workload not in STEP_KERNELS:
        return None
    ents = []
    for pre, suf in STEP_KERNELS[workload]:
        c = [e for e in tab.values() if e["kernel"].startswith(pre) and e["kernel"].endswith(suf) and e.get("avg_us")]
        if not c:
            return None
        ents.append(max(c, key=lambda e: e.get("calls", 0)))
    alg = STEP_ALG_BYTES_PER_SAMPLE[workload] * STEP_BATCH[workload]
    out = {"kernels": [f"{e['kernel']}@{e['grid_threads']}" for e in ents], "sum_avg_us": round(sum(e["avg_us"] for e in ents), 3),
           "algorithmic_bytes": alg, "batch": STEP_BATCH[workload]}
    if all(e.get("hbm_bytes") is not None for e in ents):
        out["sum_hbm_bytes"] = sum(e["hbm_bytes"] for e in ents)
        out["traffic_ratio"] = round(out["sum_hbm_bytes"] / alg, 3)
    else:
        out["missing_counters"] = [e["kernel"] for e in ents if e.get("hbm_bytes") is None]
    return out


def main():
    tag = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/p"
    os.makedirs("profiles", exist_ok=True)
    phases = {}
    train_legs = {k: k for k in ("kt", "fetch", "write", "sq", "l2") if os.path.isdir(f"{src}/{k}")}
    phases["train"] = summarize_phase(src, "train", train_legs)
    workload = tag.split("_", 1)[1] if "_" in tag else tag
    stt = step_total(phases["train"], workload)
    pred_legs = {k: f"p{k}" for k in ("kt", "fetch", "write", "sq", "l2") if os.path.isdir(f"{src}/p{k}")}
    if pred_legs:
        phases["predict"] = summarize_phase(src, "predict", pred_legs)
    st = find(src, "kt", "_kernel_stats.csv")
    if st:
        shutil.copy(st, f"profiles/{tag}_kernel_stats.csv")
    st = find(src, "pkt", "_kernel_stats.csv")
    if st:
        shutil.copy(st, f"profiles/{tag}_predict_kernel_stats.csv")
    for f, dst in (("kt_bench.json", f"profiles/{tag}_bench_under_rocprof.json"), ("pkt_bench.json", f"profiles/{tag}_predict_bench_under_rocprof.json")):
        if os.path.exists(f"{src}/{f}") and os.path.getsize(f"{src}/{f}") > 0:
            shutil.copy(f"{src}/{f}", dst)
    try:
        head = open(f"{src}/HEAD").read().strip()
    except Exception:
        head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    doc = {"source": "rocprofv3 on bench.py, one process per pass and per phase (scripts/prof_workload.sh): --kernel-trace --stats (graph "
                     "replay as benchmarked + the eager instrumented re-run), --pmc FETCH_SIZE, --pmc WRITE_SIZE, --pmc SQ_*, --pmc "
                     "TCC_HIT_sum TCC_MISS_sum (GOCTR_NO_GRAPH=1: one dispatch record per launch).  fetch_bytes = FETCH_SIZE KiB x 1024 "
                     "x 2 (gfx950 correction, MI355X_MICROARCH.md 'HBM'); write_bytes = WRITE_SIZE KiB x 1024; hbm_bytes = their sum. "
                     "Keys: '<kernel symbol>@<grid threads>' inside the phase the pass ran.",
           "commit": head, "phases": phases}
    if stt:
        doc["step_total"] = stt
    json.dump(doc, open(f"profiles/{tag}_kernels.json", "w"), indent=1)
    with open(f"profiles/{tag}_kernel_trace.txt", "w") as f:
        f.write(f"# rocprofv3 per phase / kernel symbol / launch shape (commit {head})\n")
        for ph, tab in phases.items():
            f.write(f"## phase: {ph}\n")
            f.write(f"{'kernel':58s} {'grid':>9s} {'calls':>6s} {'avg_us':>8s} {'min_us':>8s} {'hbm_MB':>8s} {'l2hit':>6s} {'mfma%':>6s}\n")
            for key, e in sorted(tab.items(), key=lambda kv: -(kv[1].get("avg_us", 0) * kv[1].get("calls", 0))):
                f.write(f"{e['kernel'][:58]:58s} {e['grid_threads']:9d} {e.get('calls', 0):6d} {e.get('avg_us', 0):8.2f} {e.get('min_us', 0):8.2f} "
                        f"{(e.get('hbm_bytes') or 0) / 1e6:8.2f} {e.get('l2_hit_rate') if e.get('l2_hit_rate') is not None else '':>6} "
                        f"{(e.get('sq') or {}).get('mfma_busy_pct', ''):>6}\n")
        if stt:
            f.write(f"## replayed training step: {' + '.join(stt['kernels'])}\n   sum of kernel durations {stt['sum_avg_us']} us; memory-side "
                    f"{stt.get('sum_hbm_bytes', 0) / 1e6:.1f} MB vs {stt['algorithmic_bytes'] / 1e6:.1f} MB algorithmic = {stt.get('traffic_ratio')} x\n")
    print(open(f"profiles/{tag}_kernel_trace.txt").read())


if __name__ == "__main__":
    main()
