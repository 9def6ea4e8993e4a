#!/bin/bash
# round 6, session 51: MLP prefetch blocks -- which cache do they fill?  GOCTR_MLP_PREFETCH=1 (shipped: rows into the readers' XCD),
# =5 (the same rows requested from the WRONG XCD: what is left is the memory-side cache's share), =3 (+ the float64 image rows for the
# weight-gradient launch), =0
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s51; mkdir -p $O
GOCTR_MLP_PREFETCH=3 timeout 900 python -m pytest tests/test_gpu_mlp.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d.get('us_per_update'), d.get('timed_regions_ms'))
P
}
for rep in 1 2 3; do
for m in 0 1 5 3; do
run mlp_p${m}_$rep "--workload mlp" GOCTR_MLP_PREFETCH=$m
done
done
for m in 1 3; do run mlp100k_p$m "--workload mlp100k" GOCTR_MLP_PREFETCH=$m; done
