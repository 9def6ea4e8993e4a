#!/bin/bash
# round 6, session 32: ctr_fwd4 with F0's two H1 tiles one after the other and the first tile's epilogue under the second tile's MFMAs:
# tests, stamps, A/B against the previous library
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s32; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_ctr.py tests/test_gpu_rank.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4
GOCTR_DBG=chain timeout 300 python scripts/ubench/fwd_phases.py 2>&1 | grep fwd4 | tail -3
GOCTR_LIB=$R/goctr_amd/libgoctr_hip_prev.so GOCTR_DBG=chain timeout 300 python scripts/ubench/fwd_phases.py 2>&1 | grep fwd4 | tail -3
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 100 --warmup 20 --no-cpu-baseline --no-serving --no-roofline --phase predict > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', 'qps', d.get('recommend_qps'))
P
}
for rep in 1 2 3; do
run din_prev_$rep "" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_prev.so
run din_new_$rep ""
run yt_prev_$rep "--workload youtube" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_prev.so
run yt_new_$rep "--workload youtube"
done
