#!/bin/bash
# round 6, session 56: mlp_tn64_kernel -- the empty k-steps of a slab's last chunk skipped; three builds side by side on one box:
# libgoctr_hip_old.so (HEAD before), libgoctr_hip.so (skip, one workgroup per CU budget), libgoctr_hip_lb2.so (skip, launch bounds (256, 2))
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s56; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlp.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d.get('us_per_update'), d.get('timed_regions_ms')[1:6])
P
}
for rep in 1 2 3; do
run mlp_old_$rep "--workload mlp" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
run mlp_new_$rep "--workload mlp"
run mlp_lb2_$rep "--workload mlp" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_lb2.so
done
run mlp100k_old "--workload mlp100k" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
run mlp100k_new "--workload mlp100k"
run mlp100k_lb2 "--workload mlp100k" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_lb2.so
