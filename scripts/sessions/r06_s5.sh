#!/bin/bash
# round 6, session 5: which of the chain's tensors should be stored through the L2 (early ones only?), with the dW slabs through it
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s5; mkdir -p $O
run() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/din_$n.json 2> $O/din_$n.err
  python - <<P
import json
d=json.loads(open('$O/din_$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], 'noPreload', (d.get('without_preload') or {}).get('ms_per_step'), 'qps', d.get('recommend_qps'), {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
}
for rep in 1 2; do
run base_$rep GOCTR_CHAIN_WT=0 GOCTR_TN_WT=0
run tn2_$rep GOCTR_CHAIN_WT=0 GOCTR_TN_WT=2
run tn2_c1_$rep GOCTR_CHAIN_WT=1 GOCTR_TN_WT=2
run tn2_c3_$rep GOCTR_CHAIN_WT=3 GOCTR_TN_WT=2
run tn2_c7_$rep GOCTR_CHAIN_WT=7 GOCTR_TN_WT=2
run tn2_c31_$rep GOCTR_CHAIN_WT=31 GOCTR_TN_WT=2
done
GOCTR_TN_WT=2 timeout 300 python bench.py --workload youtube --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/yt_tn2.json 2> $O/yt_tn2.err
GOCTR_TN_WT=0 timeout 300 python bench.py --workload youtube --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/yt_tn0.json 2> $O/yt_tn0.err
python - <<P
import json
for x in (0,2):
    d=json.loads(open('$O/yt_tn%d.json'%x).read().strip().splitlines()[-1]); print('youtube tn_wt=%d'%x, d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
