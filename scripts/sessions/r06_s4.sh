#!/bin/bash
# round 6, session 4: stores THROUGH the L2 (sc1) so that a launch leaves nothing dirty for its boundary: chain / dW / attention A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s4; mkdir -p $O
run() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/din_$n.json 2> $O/din_$n.err
  python - <<P
import json
d=json.loads(open('$O/din_$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], 'noPreload', (d.get('without_preload') or {}).get('ms_per_step'), 'qps', d.get('recommend_qps'), {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
}
for rep in 1 2; do
run base_$rep GOCTR_CHAIN_WT=0 GOCTR_TN_WT=0 GOCTR_ATTN_WT=0
run chain_$rep GOCTR_CHAIN_WT=1 GOCTR_TN_WT=0 GOCTR_ATTN_WT=0
run chain_tn1_$rep GOCTR_CHAIN_WT=1 GOCTR_TN_WT=1 GOCTR_ATTN_WT=0
run chain_tn2_$rep GOCTR_CHAIN_WT=1 GOCTR_TN_WT=2 GOCTR_ATTN_WT=0
run all_$rep GOCTR_CHAIN_WT=1 GOCTR_TN_WT=2 GOCTR_ATTN_WT=1
run chain_attn_$rep GOCTR_CHAIN_WT=1 GOCTR_TN_WT=0 GOCTR_ATTN_WT=1
done
timeout 900 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_pipeline.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
GOCTR_TN_WT=2 GOCTR_ATTN_WT=1 timeout 900 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_pipeline.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
