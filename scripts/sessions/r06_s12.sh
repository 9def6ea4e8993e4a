#!/bin/bash
# round 6, session 12: the k-NN line from a compiled host (knn_bench) beside the Python loop; 256 queries per call; search tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s12; mkdir -p $O
for rep in 1 2; do
timeout 300 python bench.py --workload knn --steps 200 --warmup 20 --no-cpu-baseline > $O/knn_$rep.json 2> $O/knn_$rep.err
python - <<P
import json
d=json.loads(open('$O/knn_$rep.json').read().strip().splitlines()[-1]); print('knn rep $rep', d['value'], d['ms_per_step'], d['timed_regions_ms'], 'python loop', d['python_loop']['value'], d['python_loop']['ms_per_step'])
P
done
for q in 1 8 64 256 1024; do goctr_amd/host/knn_bench --queries $q --steps 200 --warmup 200 --regions 5; done 2>&1 | tee $O/knn_bench_q.txt
timeout 600 python -m pytest tests/test_gpu_search.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
