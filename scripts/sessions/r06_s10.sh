#!/bin/bash
# round 6, session 10: attention forward trimmed (ids checked once per slot lane, no slot test in the pooling, one pass of side
# features) + XCD-affine tile walk of the forward-only chain: A/B against the previous commit's library (goctr_amd/libgoctr_hip_prev.so);
# tests; mlp100k kernel trace (eager launches: rocprofv3 crashes inside the graph capture of this workload)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s10; mkdir -p $O
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], 'qps', d.get('recommend_qps'), {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
}
for rep in 1 2 3; do
run din_prev_$rep "" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_prev.so
run din_new_$rep ""
done
run yt_prev "--workload youtube" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_prev.so
run yt_new "--workload youtube"
timeout 1500 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_rank.py tests/test_gpu_multi.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -4
GOCTR_NO_GRAPH=1 PASSES=kt scripts/prof_workload.sh mlp100k --workload mlp100k --regions 1 2>&1 | tail -3
f=$(find gpurun_out/p_mlp100k/kt -name "*kernel_stats.csv" | head -1); head -12 "$f"
