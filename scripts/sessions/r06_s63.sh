#!/bin/bash
# round 6, session 63: chain launch with 2 instead of 3 samples' rows gathered under F0 (no spill: 244 registers; 3 early now spills 8 and
# drains the W0 ring at the spill), and mlp_chain_kernel's four tail-chunk loads in one round.  Libraries: libgoctr_hip.so (3 early, MLP
# fix), libgoctr_hip_e2.so (2 early, MLP fix), libgoctr_hip_old.so (3 early, MLP as committed)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s63; mkdir -p $O
GOCTR_LIB=$R/goctr_amd/libgoctr_hip_e2.so timeout 1200 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_pipeline.py tests/test_gpu_mlp.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
run() {  # name, args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], (d.get('timed_regions_ms') or [])[1:6])
P
}
for rep in 1 2 3; do
run din_e3_$rep "--steps 200 --warmup 20"
run din_e2_$rep "--steps 200 --warmup 20" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_e2.so
done
run din20_e3 "--steps 20 --warmup 5"
run din20_e2 "--steps 20 --warmup 5" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_e2.so
for rep in 1 2 3; do
run mlp_old_$rep "--workload mlp --steps 200 --warmup 20" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
run mlp_new_$rep "--workload mlp --steps 200 --warmup 20"
done
run mlp100k_old "--workload mlp100k" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
run mlp100k_new "--workload mlp100k"
