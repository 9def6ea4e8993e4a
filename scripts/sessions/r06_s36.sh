#!/bin/bash
# round 6, session 36: predict on two streams (attention of group g + 1 beside the chain of group g): correctness, then rows/s with the chain on
# one / two workgroups per CU
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s36; mkdir -p $O
GOCTR_PRED_STREAMS=2 timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_ctr.py -q -m gpu -p no:cacheprovider -k "predict or forward or full_size or grouped" 2>&1 | tail -4
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 100 --warmup 20 --no-cpu-baseline --no-serving --no-roofline --phase predict > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', 'qps', d.get('recommend_qps'))
P
}
for rep in 1 2 3; do
run din_one_$rep ""
run din_two_w2_$rep "" GOCTR_PRED_STREAMS=2
run din_two_w1_$rep "" GOCTR_PRED_STREAMS=2 GOCTR_FWD4_WGS=1
run din_two_old_$rep "" GOCTR_PRED_STREAMS=2 GOCTR_FWD4=0
done
run yt_one "--workload youtube"
run yt_two_w2 "--workload youtube" GOCTR_PRED_STREAMS=2
run yt_two_w1 "--workload youtube" GOCTR_PRED_STREAMS=2 GOCTR_FWD4_WGS=1
