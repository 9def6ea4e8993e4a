#!/bin/bash
# round 6, session 45: WRITE_SIZE / FETCH_SIZE of the step's last launch on ONE box: the library of commit b293275 (profiled at 9.9 MB written), HEAD
# with the reduce block + flag (GOCTR_ATT0_EARLY=0), HEAD default
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_s45; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name, env...
  n=$1; shift
  for C in WRITE_SIZE FETCH_SIZE; do
  env "$@" GOCTR_NO_GRAPH=1 GOCTR_EAGER_PIPELINE=1 timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/${n}_$C -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-serving --no-roofline --phase train > $O/${n}_$C.json 2> $O/${n}_$C.err
  python - <<P
import csv,glob,collections
f=glob.glob('$O/${n}_$C/*/*_counter_collection.csv')[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name']=='$C': acc[r['Kernel_Name'][12:50]].append(float(r['Counter_Value']))
for k,v in acc.items():
    if any(x in k for x in ('chain_x3','x3w','reduce_attn')): print('  $n $C KiB', k, len(v), round(sum(v)/len(v),1), 'min', min(v), 'max', max(v))
P
  done
}
run old GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
run head_flag GOCTR_ATT0_EARLY=0
run head
run old2 GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
find $O -type f ! -name '*.json' ! -name '*.err' -delete
