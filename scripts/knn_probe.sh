#!/bin/bash
# k-NN search: parity tests, bench line, per-kernel durations (rocprofv3 --kernel-trace --stats) of the scan path and the tile path
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R && timeout 900 python -m pytest tests/test_gpu_search.py -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for S in ${KNN_VARIANTS:-1 0}; do
  echo "== GOCTR_KNN_SCAN=$S"
  GOCTR_KNN_SCAN=$S python $R/bench.py --workload knn --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   queries/s', d['value'], 'ms/call', d['ms_per_step'])"
  rm -rf /tmp/kk
  GOCTR_KNN_SCAN=$S rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kk -- python $R/bench.py --workload knn --no-cpu-baseline > /dev/null 2>&1
  python3 - <<'PY'
import csv,glob
for f in glob.glob("/tmp/kk/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        print("   %-44s calls %4s avg %9.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:44], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
