#!/bin/bash
# round 6, session 53: sklearn-port MLP with the float64 row image + prefetch blocks as shipped: tests, rocprofv3 passes (mlp, mlp100k), lines
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s53; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_fullsize.py -q -m gpu -x -p no:cacheprovider -k "mlp or Mlp or flagship or sklearn" 2>&1 | tail -3
KT_EAGER=1 PASS_TIMEOUT=240 scripts/prof_workload.sh mlp --workload mlp > $O/prof_mlp.log 2>&1; tail -3 $O/prof_mlp.log
GOCTR_NO_GRAPH=1 PASSES=kt PASS_TIMEOUT=240 scripts/prof_workload.sh mlp100k --workload mlp100k --regions 1 > $O/prof_mlp100k.log 2>&1; tail -3 $O/prof_mlp100k.log
mkdir -p gpurun_out/bench
timeout 300 python bench.py --workload mlp > gpurun_out/bench/mlp.json 2> gpurun_out/bench/mlp.err
timeout 300 python bench.py --workload mlp100k > gpurun_out/bench/mlp100k.json 2> gpurun_out/bench/mlp100k.err
for f in gpurun_out/bench/mlp.json gpurun_out/bench/mlp100k.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['unit'], 'ms/step', d.get('ms_per_step'), d.get('us_per_update'), 'roofline', (d.get('roofline') or {}).get('frac'))
"; done
