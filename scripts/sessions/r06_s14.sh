#!/bin/bash
# round 6, session 14: the attention forward as the chain launch's head (GOCTR_CHAIN_HEAD=1): bit-equality with the
# separate launches, A/B, stamps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -x -p no:cacheprovider -k "pipelined_graphs_equal" 2>&1 | tail -4
GOCTR_CHAIN_HEAD=1 timeout 900 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_fullsize.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
run() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/din_$n.json 2> $O/din_$n.err
  python - <<P
import json
d=json.loads(open('$O/din_$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], 'noPreload', (d.get('without_preload') or {}).get('ms_per_step'), 'qps', d.get('recommend_qps'), {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
}
for rep in 1 2 3; do
run head0_$rep GOCTR_CHAIN_HEAD=0
run head1_$rep GOCTR_CHAIN_HEAD=1
done
for x in 0 1; do echo "head=$x"; GOCTR_CHAIN_HEAD=$x timeout 120 python scripts/dbg_chain.py 2>&1 | grep -h "phases" | tail -3; done
