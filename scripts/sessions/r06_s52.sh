#!/bin/bash
# round 6, session 52: MLP prefetch blocks -- the XCD a prefetch block should serve: all eight shifts of (blockIdx + shift) % 8
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s52; mkdir -p $O
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d.get('us_per_update'), d.get('timed_regions_ms')[1:6])
P
}
for rep in 1 2; do
for sh in 0 1 2 3 4 5 6 7; do
run mlp_sh${sh}_$rep "--workload mlp" GOCTR_MLP_PREFETCH=3 GOCTR_MLP_PF_SHIFT=$sh
done
done
for sh in 0 2 4 6; do run mlp100k_sh$sh "--workload mlp100k" GOCTR_MLP_PREFETCH=3 GOCTR_MLP_PF_SHIFT=$sh; done
