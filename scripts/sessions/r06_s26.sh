#!/bin/bash
# round 6, session 26: ctr_fwd4 variants (ring depths, rotating output wavefront, no stamps) -- recommend rows/s, 3 repeats each
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s26; mkdir -p $O
run() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-serving --no-roofline --phase predict > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d.get('recommend_qps'))
P
}
for rep in 1 2 3; do
run off_$rep GOCTR_FWD4=0
run base_$rep
for v in v_rot v_r3 v_nodbg v_nodbgrot; do run ${v}_$rep GOCTR_LIB=$R/goctr_amd/libgoctr_hip_$v.so; done
done
