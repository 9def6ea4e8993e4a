#!/bin/bash
# round 5, session 19: bar_alloc() behind the /proc/self/maps check -- is the BAR path still taken?  (k-NN line, rank under 8 callers)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s19; rm -rf $O; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_gpu_rank.py tests/test_gpu_search.py -m gpu -q 2>&1 | tail -4) > $O/pytest.log; tail -2 $O/pytest.log
for g in 1 0 1; do
  GOCTR_KNN_BAR=$g timeout 200 python bench.py --workload knn --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('knn bar $g', d['value'], d['ms_per_step'])"
  GOCTR_SERVE_BAR=$g timeout 120 goctr_amd/host/rank_bench --threads 8 --n 256 --seconds 0.3 --kind din --coalesce 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('rank bar $g', [(e['n'],e['threads'],e['latency_us']['p50'],e['latency_us']['p99']) for e in d['results']])"
done | tee $O/out.txt
