#!/bin/bash
# round 5, session 14: does a pre-load in TRAINING kernels (scratch model, GOCTR_BENCH_PRELOAD_STEPS) flatten the nine regions of a
# --steps 20 run further than the predict leg alone?  Interleaved.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s14; rm -rf $O; mkdir -p $O
cd $R
for rep in 1 2 3; do for g in 0 2000 8000; do
  GOCTR_BENCH_PRELOAD_STEPS=$g timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serving --no-roofline 2>/dev/null | tail -1 > $O/din_p${g}_rep$rep.json
done; done
python3 - <<PY | tee $O/summary.txt
import json
for g in (0,2000,8000):
    for i in (1,2,3):
        d=json.loads(open("$O/din_p%d_rep%d.json"%(g,i)).read())
        print("preload",g,"rep",i,d["value"],d["timed_regions_ms"])
PY
