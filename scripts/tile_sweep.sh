#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for B in 4096 8192 16384 32768; do
  rm -rf /tmp/ts_$B
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts_$B -- python $R/scripts/tile_sweep.py $B 2>/dev/null | grep "us/step"
  f=$(find /tmp/ts_$B -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,re
for row in csv.reader(open(sys.argv[1])):
    if row and re.search(r'chain_x3_kernel|tn_multi_x3w|reduce_attn', row[0]):
        print('   %-40s calls %5s avg %8.2f us' % (re.search(r'(ctr_chain_x3_kernel<[^>]*>|gemm_tn_multi_x3w_kernel<[^>]*>|reduce_attn_kernel<[^>]*>)', row[0]).group(1), row[1], float(row[3])/1e3))
PY
done
