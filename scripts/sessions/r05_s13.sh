#!/bin/bash
# round 5, session 13: release-only system fences (no L2 invalidate) in front of the completion words of the k-NN collect kernel
# and the one-launch serving pass: polling forced for every size against the stream wait
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s13; rm -rf $O; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_gpu_rank.py tests/test_gpu_search.py -m gpu -q 2>&1 | tail -6) > $O/pytest.log; tail -3 $O/pytest.log
for rep in 1 2; do for g in 100000 0; do
  GOCTR_KNN_POLL_MAXQ=$g KNN_LATENCY_SCAN=1 KNN_LATENCY_Q=1,8,32,64,128,256 timeout 120 python scripts/knn_latency.py 2>/dev/null | sed "s/^/poll_maxq $g /"
done; done | tee $O/knn_latency.txt
for rep in 1 2; do for g in 100000 0; do
  GOCTR_SERVE_POLL_ROWS=$g timeout 120 goctr_amd/host/rank_bench --threads 1,8 --n 32,256,512,1024,2048 --seconds 0.3 --kind din --coalesce 1 2>/dev/null | tail -1 > $O/rank_p${g}_rep$rep.json
done; done
python3 - <<PY | tee $O/rank.txt
import json
for g in (100000,0):
    for i in (1,2):
        d=json.loads(open("$O/rank_p%d_rep%d.json"%(g,i)).read())
        print("poll",g,"rep",i," ".join("n%d_t%d %.1f/%.1f"%(e["n"],e["threads"],e["latency_us"]["p50"],e["latency_us"]["p99"]) for e in d["results"]), d.get("bit_equal_to_single_threaded"))
PY
