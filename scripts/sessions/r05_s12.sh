#!/bin/bash
# round 5, session 12: the one-launch serving pass -- the host watching the workgroups' stamps in the pinned buffer (GOCTR_SERVE_POLL_ROWS=100000)
# against the stream wait (GOCTR_SERVE_POLL_ROWS=0): rank / serving tests, then goctr_amd/host/rank_bench interleaved
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s12; rm -rf $O; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_gpu_rank.py tests/test_gpu_assembly.py -m gpu -q 2>&1 | tail -6) > $O/pytest.log; tail -3 $O/pytest.log
for rep in 1 2; do for g in 100000 0 256; do
  GOCTR_SERVE_POLL_ROWS=$g timeout 120 goctr_amd/host/rank_bench --threads 1,8 --n 32,256,512,1024,2048 --seconds 0.3 --kind din --coalesce 1 2>/dev/null | tail -1 > $O/rank_p${g}_rep$rep.json
done; done
python3 - <<PY
import json
for g in (100000,0,256):
    for i in (1,2):
        d=json.loads(open("$O/rank_p%d_rep%d.json"%(g,i)).read())
        print("poll",g,"rep",i," ".join("n%d_t%d p50 %.1f p99 %.1f"%(e["n"],e["threads"],e["latency_us"]["p50"],e["latency_us"]["p99"]) for e in d["results"]), d.get("bit_equal_to_single_threaded"))
PY
