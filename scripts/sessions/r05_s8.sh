#!/bin/bash
# round 5, GPU session 8: the bf16-plane k-NN filter -- bit-exact tests, then A/B + kernel times
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s8; rm -rf $O; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_search.py -m gpu -q -x 2>&1 | tail -15) > $O/pytest.log
tail -5 $O/pytest.log
timeout 200 python bench.py --workload knn --no-cpu-baseline > $O/knn_bf16.json 2> $O/knn_bf16.err
GOCTR_KNN_BF16=0 timeout 200 python bench.py --workload knn --no-cpu-baseline > $O/knn_f32mfma.json 2> $O/knn_f32mfma.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --workload knn --no-cpu-baseline > $O/kt.json 2> $O/kt.err
python3 - <<PY
import csv,glob,json
for f in glob.glob("$O/kt/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("  ", r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,2))
for n in ("knn_bf16","knn_f32mfma"):
    d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"])
PY
find $O/kt -type f ! -name '*_kernel_stats.csv' -delete
