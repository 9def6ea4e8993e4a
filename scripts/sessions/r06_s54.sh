#!/bin/bash
# round 6, session 54: sklearn-port MLP -- weight-gradient workgroups dealt XCD-affine (GOCTR_MLP_TN_XCD): tests, A/B, kernel trace
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s54; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_fullsize.py -q -m gpu -x -p no:cacheprovider -k "mlp or Mlp or flagship or sklearn" 2>&1 | tail -3
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d.get('us_per_update'), d.get('timed_regions_ms')[1:6])
P
}
for rep in 1 2 3; do
for m in 0 1; do run mlp_t${m}_$rep "--workload mlp" GOCTR_MLP_TN_XCD=$m; done
done
for m in 0 1; do run mlp100k_t$m "--workload mlp100k" GOCTR_MLP_TN_XCD=$m; done
KT_EAGER=1 PASSES="kt fetch l2" PASS_TIMEOUT=240 scripts/prof_workload.sh mlp --workload mlp > $O/prof_mlp.log 2>&1; tail -2 $O/prof_mlp.log
