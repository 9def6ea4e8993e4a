#!/bin/bash
# round 5, session 20b: with the bf16 matrix-core scan forced (GOCTR_KNN_MFMA=1), does polling the completion words still pay at
# 12..47 queries per call?  GOCTR_KNN_POLL_MAXQ = 0 (stream wait) / 100000 (poll)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s20; mkdir -p $O
cd $R
export GOCTR_KNN_MFMA=1
for rep in 1 2; do for g in 0 100000; do
  GOCTR_KNN_POLL_MAXQ=$g KNN_LATENCY_SCAN=1 KNN_LATENCY_Q=40,48,56,64,96,128 timeout 120 python scripts/knn_latency.py 2>/dev/null | sed "s/^/poll $g /"
done; done | tee $O/latency_poll2.txt
