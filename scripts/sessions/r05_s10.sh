#!/bin/bash
# round 5, session 10: k-NN call completion -- the host polling the pinned counts (calls of up to GOCTR_KNN_POLL_MAXQ queries)
# against hipStreamSynchronize, by queries per call
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s10; rm -rf $O; mkdir -p $O
cd $R
(timeout 500 python -m pytest tests/test_gpu_search.py -m gpu -q 2>&1 | tail -6) > $O/pytest.log; tail -3 $O/pytest.log
(GOCTR_KNN_POLL_MAXQ=100000 timeout 500 python -m pytest tests/test_gpu_search.py -m gpu -q 2>&1 | tail -6) > $O/pytest_poll.log; tail -3 $O/pytest_poll.log
for rep in 1 2; do for g in 100000 0; do
  GOCTR_KNN_POLL_MAXQ=$g KNN_LATENCY_SCAN=1 KNN_LATENCY_Q=1,2,4,8,16,32,64,128 timeout 120 python scripts/knn_latency.py 2>/dev/null | sed "s/^/poll_maxq $g /"
done; done | tee $O/latency.txt
for rep in 1 2 3; do
  timeout 200 python bench.py --workload knn --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench default', d['value'], d['ms_per_step'])"
done | tee $O/bench.txt
