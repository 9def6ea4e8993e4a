#!/bin/bash
# round 5, GPU session 9: merged MLP gradient + update launch -- parity, then A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s9; rm -rf $O; mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_gpu_mlp.py -m gpu -q -x 2>&1 | tail -12) > $O/pytest.log; tail -5 $O/pytest.log
GOCTR_MLP_MERGE=1 timeout 200 python bench.py --workload mlp --no-cpu-baseline > $O/merge1.json 2> $O/merge1.err
GOCTR_MLP_MERGE=0 timeout 200 python bench.py --workload mlp --no-cpu-baseline > $O/merge0.json 2> $O/merge0.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --workload mlp --no-cpu-baseline > $O/kt.json 2> $O/kt.err
python3 - <<PY
import csv,glob,json
for f in glob.glob("$O/kt/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("  ", r["Name"][:80], r["Calls"], round(float(r["AverageNs"])/1e3,2))
for n in ("merge1","merge0"):
    d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"])
PY
find $O/kt -type f ! -name '*_kernel_stats.csv' -delete
tail -3 $O/merge1.err
