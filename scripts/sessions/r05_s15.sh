#!/bin/bash
# round 5, session 15: the default bench line with the training pre-load (every leg on), the driver's flags, YouTube and --train-emb
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s15; rm -rf $O; mkdir -p $O
cd $R
timeout 400 python bench.py > $O/din.json 2> $O/din.err
timeout 300 python bench.py --steps 20 --warmup 5 > $O/din_steps20.json 2> $O/din_steps20.err
timeout 400 python bench.py --workload youtube --no-cpu-baseline > $O/youtube.json 2> $O/youtube.err
timeout 300 python bench.py --train-emb 0.05 --no-cpu-baseline > $O/din_trainemb.json 2> $O/din_trainemb.err
python3 - <<PY
import json
for f in ("din","din_steps20","youtube","din_trainemb"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["timed_regions_ms"], d["preload"]["scratch_model_training_steps"], d.get("recommend_qps"))
    except Exception as e: print(f, "ERR", e)
PY
for f in $O/*.err; do tail -n 3 $f; done
