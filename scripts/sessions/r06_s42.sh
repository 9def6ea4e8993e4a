#!/bin/bash
# round 6, session 42: what the attention wavefronts' wait for the att0 flag costs the step's last launch -- a TIMING experiment
# (GOCTR_EXP_NOWAIT=1: no wait, the attention reads the previous step's att0: wrong results, same work)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s42; mkdir -p $O
train() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], {k:v.get('avg_us') for k,v in d.get('kernels',{}).items() if isinstance(v,dict)})
P
}
for rep in 1 2 3; do
train wait_$rep
train nowait_$rep GOCTR_EXP_NOWAIT=1
done
