#!/bin/bash
# round 6, session 31: ctr_fwd4 against the 8-wavefront kernel at smaller launches (GOCTR_PRED_GROUP = 2 / 4: 256 / 512 tiles per launch)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s31; mkdir -p $O
run() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-serving --no-roofline --phase predict > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d.get('recommend_qps'))
P
}
for rep in 1 2; do
for g in 2 4 8; do
run g${g}_off_$rep GOCTR_PRED_GROUP=$g GOCTR_FWD4=0
run g${g}_new_$rep GOCTR_PRED_GROUP=$g
done
done
