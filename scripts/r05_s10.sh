#!/bin/bash
# round 5, session 10: the k-NN call as one graph launch (upload + filter + collect) against three stream commands
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s10; rm -rf $O; mkdir -p $O
cd $R
(timeout 500 python -m pytest tests/test_gpu_search.py -m gpu -q 2>&1 | tail -6) > $O/pytest.log; tail -3 $O/pytest.log
for rep in 1 2 3; do for g in 1 0; do
  GOCTR_KNN_GRAPH=$g timeout 200 python bench.py --workload knn --no-cpu-baseline --no-roofline > $O/knn_g${g}_rep$rep.json 2>/dev/null
done; done
python3 - <<PY
import json
for g in (1,0):
    for i in (1,2,3):
        d=json.loads(open("$O/knn_g%d_rep%d.json"%(g,i)).read().strip().splitlines()[-1]); print("graph",g,"rep",i,d["value"],d["ms_per_step"],d.get("timed_region_spread"))
PY
