#!/bin/bash
# round 5, final GPU session: full -m gpu suite, smoke, every bench line, rocprofv3 passes (scripts/prof_workload.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/pytest.log; tail -4 $O/pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $O/smoke.log; cat $O/smoke.log
scripts/bench_round.sh > $O/bench_round.log 2>&1; tail -30 $O/bench_round.log
for i in 1 2 3; do timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serving --no-roofline > $O/din20_rep$i.json 2>/dev/null; done
python3 - <<PY
import json
for i in (1,2,3):
    d=json.loads(open("$O/din20_rep%d.json"%i).read().strip().splitlines()[-1]); print("din20 rep",i,d["value"],d["ms_per_step"],d["timed_region_spread"])
PY
PREDICT=1 scripts/prof_workload.sh din > $O/prof_din.log 2>&1
scripts/prof_workload.sh knn --workload knn > $O/prof_knn.log 2>&1
scripts/prof_workload.sh item2vec --workload item2vec > $O/prof_item2vec.log 2>&1
scripts/prof_workload.sh mlp --workload mlp > $O/prof_mlp.log 2>&1
PREDICT=1 scripts/prof_workload.sh youtube --workload youtube > $O/prof_youtube.log 2>&1
du -sh $R/gpurun_out/p_* | tail -8
