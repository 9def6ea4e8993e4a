#!/bin/bash
# round 5, session 18: the serving pass's keys stored by the host into device memory over the PCIe BAR (GOCTR_SERVE_BAR=1, default)
# against the kernels reading the pinned host buffer (0): rank / assembly tests, then goctr_amd/host/rank_bench interleaved
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s18; rm -rf $O; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_gpu_rank.py tests/test_gpu_assembly.py -m gpu -q 2>&1 | tail -6) > $O/pytest.log; tail -3 $O/pytest.log
for rep in 1 2 3; do for g in 1 0; do
  GOCTR_SERVE_BAR=$g timeout 120 goctr_amd/host/rank_bench --threads 1,8 --n 32,256,1024,2048 --seconds 0.3 --kind din --coalesce 1 2>/dev/null | tail -1 > $O/rank_b${g}_rep$rep.json
done; done
python3 - <<PY | tee $O/rank.txt
import json
for g in (1,0):
    for i in (1,2,3):
        d=json.loads(open("$O/rank_b%d_rep%d.json"%(g,i)).read())
        print("bar",g,"rep",i," ".join("n%d_t%d %.1f/%.1f"%(e["n"],e["threads"],e["latency_us"]["p50"],e["latency_us"]["p99"]) for e in d["results"]), d.get("bit_equal_to_single_threaded"))
PY
