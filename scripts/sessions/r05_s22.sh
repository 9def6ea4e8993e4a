#!/bin/bash
# round 5, session 22: the k-NN thresholds after the BAR input -- matrix-core scan from 12 queries per call on, completion polled up to
# 64 queries; the float32 MFMA kernel left the tree: search tests, latency by queries per call, bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s22; rm -rf $O; mkdir -p $O
cd $R
(timeout 500 python -m pytest tests/test_gpu_search.py -m gpu -q 2>&1 | tail -6) > $O/pytest.log; tail -3 $O/pytest.log
for rep in 1 2; do KNN_LATENCY_SCAN=1 KNN_LATENCY_Q=1,8,12,16,32,47,64,96,128,256 timeout 120 python scripts/knn_latency.py 2>/dev/null; done | tee $O/latency.txt
for rep in 1 2 3; do timeout 200 python bench.py --workload knn --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"; done | tee $O/bench.txt
