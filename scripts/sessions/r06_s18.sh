#!/bin/bash
# round 6, session 18: sklearn-port MLP -- mlp_reduce_update_kernel requests W and the moments in front of the slab sums: tests + A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_multi.py -q -m gpu -x -p no:cacheprovider -k "mlp or Mlp or sklearn" 2>&1 | tail -3
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d.get('us_per_update'))
P
}
for rep in 1 2 3; do
run mlp_prev_$rep "--workload mlp" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_prev.so
run mlp_new_$rep "--workload mlp"
done
run mlp100k_prev "--workload mlp100k" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_prev.so
run mlp100k_new "--workload mlp100k"
