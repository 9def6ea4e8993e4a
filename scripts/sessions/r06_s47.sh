#!/bin/bash
# round 6, session 47: the chain launch's per-tile att0 sums stored through the L2 (the att0 workgroup of the weight-gradient launch waits for them):
# rocprofv3 durations and the bench line against the previous build (libgoctr_hip_prev.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_s47; mkdir -p $O
cd $R
train() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'])
P
}
for rep in 1 2 3 4; do
train new_$rep
train prev_$rep GOCTR_LIB=$R/goctr_amd/libgoctr_hip_prev.so
done
cd /tmp && export TMPDIR=/tmp
for L in libgoctr_hip.so libgoctr_hip_prev.so; do
  for rep in 1 2; do
  GOCTR_LIB=$R/goctr_amd/$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${L}_$rep -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving --phase train > $O/kt_${L}_$rep.json 2> $O/kt_${L}_$rep.err
  echo "== $L rep $rep"; python - <<P
import csv,glob
f=glob.glob('$O/kt_${L}_$rep/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if any(k in n for k in ('chain_x3','x3w','reduce_attn')): print('  ', n[12:52], r['Calls'], round(float(r['AverageNs'])/1000,2))
P
  done
done
find $O -type f ! -name '*.json' ! -name '*.err' -delete
