#!/bin/bash
# round 6, session 19: forward-only chain reading the static h0 columns from their sources (attention launch writes pooled columns only):
# tests, A/B against GOCTR_PRED_DIRECT=0 and the previous library, forward-only phase stamps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s19; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ctr.py tests/test_gpu_fullsize.py tests/test_gpu_rank.py tests/test_gpu_model_e2e.py -q -m gpu -p no:cacheprovider 2>&1 | tail -6
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline --no-serving --no-roofline > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], 'qps', d.get('recommend_qps'))
P
}
for rep in 1 2 3; do
run din_prev_$rep "" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_prev.so
run din_off_$rep "" GOCTR_PRED_DIRECT=0
run din_new_$rep ""
done
run yt_off "--workload youtube" GOCTR_PRED_DIRECT=0
run yt_new "--workload youtube"
GOCTR_DBG=chain timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serving --no-roofline 2>&1 | grep "forward-only phases" | tail -3
GOCTR_DBG=chain GOCTR_PRED_DIRECT=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serving --no-roofline 2>&1 | grep "forward-only phases" | tail -3
