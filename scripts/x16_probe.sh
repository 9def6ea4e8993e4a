#!/bin/bash
# chain kernel A/B: 16-row (GOCTR_CHAIN_X16=1) vs 32-row tiles at cfg3: scripts/gpu.sh -- scripts/x16_probe.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for X in 1 0; do
  echo "== GOCTR_CHAIN_X16=$X"
  GOCTR_CHAIN_X16=$X GOCTR_NO_GRAPH=1 GOCTR_EAGER_PIPELINE=1 GOCTR_CHAIN_DBG=1 timeout 100 python $R/scripts/tile_sweep.py 8192 2>&1 | grep "phases" | tail -3
  rm -rf /tmp/x16_$X
  GOCTR_CHAIN_X16=$X rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/x16_$X -- python $R/scripts/tile_sweep.py 8192 2>/dev/null | grep "us/step"
  f=$(find /tmp/x16_$X -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,re
for row in csv.reader(open(sys.argv[1])):
    if row and re.search(r'chain_x|tn_multi_x3w|reduce_attn', row[0]):
        print('   %-40s calls %5s avg %8.2f us' % (re.search(r'(ctr_chain_x\d+_kernel<[^>]*>|gemm_tn_multi_x3w_kernel<[^>]*>|reduce_attn_kernel<[^>]*>)', row[0]).group(1), row[1], float(row[3])/1e3))
PY
done
