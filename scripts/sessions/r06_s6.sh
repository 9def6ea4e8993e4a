#!/bin/bash
# round 6, session 6: early L2 write-back requests inside the chain launch (one workgroup per XCD, no wait); e2e drift probe; mlp100k line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s6; mkdir -p $O
run() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/din_$n.json 2> $O/din_$n.err
  python - <<P
import json
d=json.loads(open('$O/din_$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], 'noPreload', (d.get('without_preload') or {}).get('ms_per_step'), 'qps', d.get('recommend_qps'), {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
}
for rep in 1 2; do
run wb0_$rep GOCTR_CHAIN_WB=0
run wb1_$rep GOCTR_CHAIN_WB=1
run wb2_$rep GOCTR_CHAIN_WB=2
run wb3_$rep GOCTR_CHAIN_WB=3
done
timeout 600 python scripts/e2e_model_test_probe.py > $O/e2e_probe.txt 2>&1; cat $O/e2e_probe.txt | tail -8
timeout 300 python bench.py --workload mlp100k > $O/mlp100k.json 2> $O/mlp100k.err; tail -c 2500 $O/mlp100k.json; tail -3 $O/mlp100k.err
timeout 600 python -m pytest tests/test_gpu_mlp.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
