#!/bin/bash
# round 6, session 9: non-temporal stores of A0 / dz0 in the chain launch (A/B); the k-NN line with and without CPython's collector
# in the timed regions; rocprofv3 kernel trace of the mlp100k fit
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s9; mkdir -p $O
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d.get('timed_regions_ms'), {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
}
for rep in 1 2; do
run din_nt0_$rep "" GOCTR_CHAIN_NT=0
run din_nt1_$rep "" GOCTR_CHAIN_NT=1
run din_nt2_$rep "" GOCTR_CHAIN_NT=2
run din_nt3_$rep "" GOCTR_CHAIN_NT=3
done
for rep in 1 2; do
run knn_gc1_$rep "--workload knn" GOCTR_BENCH_GC=1
run knn_gc0_$rep "--workload knn" GOCTR_BENCH_GC=0
done
PASSES=kt scripts/prof_workload.sh mlp100k --workload mlp100k --regions 1 2>&1 | tail -3
f=$(find gpurun_out/p_mlp100k/kt -name "*kernel_stats.csv" | head -1); head -12 "$f"
timeout 300 python -m pytest tests/test_gpu_ctr.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
