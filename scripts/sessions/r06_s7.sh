#!/bin/bash
# round 6, session 7: per-tile sums of dW2 / att0 out of the chain launch (A/B + tests); write-through slabs for the f64 MLP and the
# k-NN scan maxima (A/B); the e2e model_test.go test
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s7; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_model_e2e.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -5
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], 'noPreload', (d.get('without_preload') or {}).get('ms_per_step'), 'qps', d.get('recommend_qps'), {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
}
for rep in 1 2; do
run din_sums0_$rep "" GOCTR_CHAIN_TILE_SUMS=0
run din_sums1_$rep "" GOCTR_CHAIN_TILE_SUMS=1
done
run yt_sums0 "--workload youtube" GOCTR_CHAIN_TILE_SUMS=0
run yt_sums1 "--workload youtube" GOCTR_CHAIN_TILE_SUMS=1
for rep in 1 2; do
run mlp_wt0_$rep "--workload mlp" GOCTR_MLP_TN_WT=0
run mlp_wt1_$rep "--workload mlp" GOCTR_MLP_TN_WT=1
run knn_wt0_$rep "--workload knn" GOCTR_KNN_WT=0
run knn_wt1_$rep "--workload knn" GOCTR_KNN_WT=1
done
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_search.py tests/test_gpu_multi.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -5
