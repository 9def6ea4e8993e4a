#!/bin/bash
# round 6, session 33: ctr_fwd4 epilogue placement variants: s1 = only the first tile's epilogue under F0's second tile; r3 / d3 / both = also
# the second tile's under F1's first chunks, with 3 W1 pieces / 3 W0 slots / both in flight (4 / 4 spills 26 registers at Ip = 144)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s33; mkdir -p $O
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
for v in v_s1 v_r3 v_d3 v_both; do run din_${v}_$rep "" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_$v.so; done
done
for rep in 1 2; do
run yt_prev_$rep "--workload youtube" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_prev.so
for v in v_s1 v_r3 v_both; do run yt_${v}_$rep "--workload youtube" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_$v.so; done
done
