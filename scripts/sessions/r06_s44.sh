#!/bin/bash
# round 6, session 44: rocprofv3 kernel durations of the replayed step with att0 updated inside the weight-gradient launch (default) and with the
# reduce block + flag (GOCTR_ATT0_EARLY=0), same box; WRITE_SIZE of the last launch both ways
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r06_s44; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for V in 1 0; do
  for rep in 1 2; do
  GOCTR_ATT0_EARLY=$V timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${V}_$rep -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving --phase train > $O/kt_${V}_$rep.json 2> $O/kt_${V}_$rep.err
  echo "== GOCTR_ATT0_EARLY=$V rep $rep"; python - <<P
import csv,glob
f=glob.glob('$O/kt_${V}_$rep/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if any(k in n for k in ('chain_x3','x3w','reduce_attn')): print('  ', n[:60], r['Calls'], r['AverageNs'])
P
  done
  GOCTR_ATT0_EARLY=$V GOCTR_NO_GRAPH=1 GOCTR_EAGER_PIPELINE=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w_$V -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-serving --no-roofline --phase train > $O/w_$V.json 2> $O/w_$V.err
  python - <<P
import csv,glob,collections
f=glob.glob('$O/w_$V/*/*_counter_collection.csv')[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name']=='WRITE_SIZE': acc[r['Kernel_Name'][:50]].append(float(r['Counter_Value']))
for k,v in acc.items():
    if any(x in k for x in ('chain_x3','x3w','reduce_attn')): print('   WRITE_SIZE KiB', k, len(v), round(sum(v)/len(v),1))
P
done
find $O -type f ! -name '*_kernel_stats.csv' ! -name '*.json' ! -name '*.err' -delete
