#!/bin/bash
# round 5, session 21: the k-NN bench line (64 queries per call, BAR input, bf16 scan) with the host polling (GOCTR_KNN_POLL_MAXQ=100)
# against the stream wait (32 = shipped so far), interleaved x4; and the scan threshold: 24 / 32 queries with the matrix-core scan
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s21; rm -rf $O; mkdir -p $O
cd $R
for rep in 1 2 3 4; do for g in 100 32; do
  GOCTR_KNN_POLL_MAXQ=$g timeout 200 python bench.py --workload knn --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench poll_maxq $g', d['value'], d['ms_per_step'])"
done; done | tee $O/bench.txt
