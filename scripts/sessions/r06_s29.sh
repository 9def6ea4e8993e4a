#!/bin/bash
# round 6, session 29: ctr_fwd4 with the exchange one H2 tile at a time (XU): YouTube (Ip = 240) on two workgroups per CU; DIN A/B of XU
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s29; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_ctr.py tests/test_gpu_rank.py -q -m gpu -p no:cacheprovider 2>&1 | tail -5
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 100 --warmup 20 --no-cpu-baseline --no-serving --no-roofline --phase predict > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', 'qps', d.get('recommend_qps'))
P
}
for rep in 1 2 3; do
run yt_off_$rep "--workload youtube" GOCTR_FWD4=0
run yt_new_$rep "--workload youtube"
run din_x0_$rep ""
run din_xu_$rep "" GOCTR_FWD4_XU=1
done
