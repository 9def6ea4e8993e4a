#!/bin/bash
# round 6, session 38: sklearn-port MLP cfg2 -- slab height of the weight-gradient launch ("halve the slab count").  The rule
# gives 98 rows (42 slabs x 6 k-blocks = 252 workgroups); GOCTR_EXP_MLP_ROWS (experiment knob, not in the tree afterwards) forces a height.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s38; mkdir -p $O
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], {k:v.get('avg_us') for k,v in d.get('kernels',{}).items() if isinstance(v,dict)})
P
}
for rep in 1 2; do
run base_$rep "--workload mlp"
for r in 64 80 128 160 196 256; do
run rows${r}_$rep "--workload mlp" GOCTR_EXP_MLP_ROWS=$r
done
done
