#!/bin/bash
# round 6, session 16: phase stamps of knn_collect_kernel (GOCTR_DBG=knn) at 64 / 1 / 256 queries per call
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s16; mkdir -p $O
for q in 64 1 256; do echo "queries per call: $q"; GOCTR_DBG=knn goctr_amd/host/knn_bench --queries $q --steps 6 --warmup 20 --regions 1 2>&1 | grep knn_collect | tail -4; done | tee $O/knn_stamps.txt
