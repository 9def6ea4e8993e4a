#!/bin/bash
# round 5, session 17: the k-NN call's input written by the host into fine-grained device memory over the PCIe BAR (GOCTR_KNN_BAR=1,
# default on large-BAR systems) against the staged hipMemcpyAsync (0): search tests, latency by queries per call, bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s17; rm -rf $O; mkdir -p $O
cd $R
(timeout 500 python -m pytest tests/test_gpu_search.py -m gpu -q 2>&1 | tail -6) > $O/pytest.log; tail -3 $O/pytest.log
for rep in 1 2; do for g in 1 0; do
  GOCTR_KNN_BAR=$g KNN_LATENCY_SCAN=1 KNN_LATENCY_Q=1,8,32,64,256 timeout 120 python scripts/knn_latency.py 2>/dev/null | sed "s/^/bar $g /"
done; done | tee $O/latency.txt
for rep in 1 2 3; do for g in 1 0; do
  GOCTR_KNN_BAR=$g timeout 200 python bench.py --workload knn --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench bar $g', d['value'], d['ms_per_step'])"
done; done | tee $O/bench.txt
