#!/bin/bash
# round 5, session 11: the shipped k-NN completion rule (host polls calls of up to 32 queries) -- search tests, latency, bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s11; rm -rf $O; mkdir -p $O
cd $R
(timeout 500 python -m pytest tests/test_gpu_search.py -m gpu -q 2>&1 | tail -6) > $O/pytest.log; tail -3 $O/pytest.log
KNN_LATENCY_SCAN=1 KNN_LATENCY_Q=1,8,32,64,256 timeout 120 python scripts/knn_latency.py 2>/dev/null | tee $O/latency.txt
timeout 200 python bench.py --workload knn > $O/bench_knn.json 2>/dev/null; tail -c 600 $O/bench_knn.json
