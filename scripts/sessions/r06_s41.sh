#!/bin/bash
# round 6, session 41: the training chain's output unit (float64 exp, two divisions) by wavefront 0 alone, d cost / d z2 handed round through LDS
# behind one more barrier, instead of by all eight wavefronts: tests, A/B against the previous build (libgoctr_hip_prev.so), stamps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s41; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
train() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py $WL --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], {k:v.get('avg_us') for k,v in d.get('kernels',{}).items() if isinstance(v,dict)})
P
}
for rep in 1 2 3 4; do
train new_$rep
train prev_$rep GOCTR_LIB=$R/goctr_amd/libgoctr_hip_prev.so
done
WL="--workload youtube" train yt_new
WL="--workload youtube" train yt_prev GOCTR_LIB=$R/goctr_amd/libgoctr_hip_prev.so
for L in libgoctr_hip.so libgoctr_hip_prev.so; do
GOCTR_LIB=$R/goctr_amd/$L timeout 120 python scripts/dbg_chain.py 2>&1 | grep "chain_x3 phases" | tail -2
done
