#!/bin/bash
# round 6, session 55: MLP prefetch blocks -- how much to request: GOCTR_MLP_PF_VAR 0 all / 1 only the first four lines of a float32 row /
# 2 image rows of each slab's first chunk only / 3 both
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s55; mkdir -p $O
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d.get('us_per_update'), d.get('timed_regions_ms')[1:6])
P
}
for rep in 1 2 3; do
for m in 0 1 2 3; do run mlp_v${m}_$rep "--workload mlp" GOCTR_MLP_PF_VAR=$m; done
done
for m in 0 1 2 3; do run mlp100k_v$m "--workload mlp100k" GOCTR_MLP_PF_VAR=$m; done
