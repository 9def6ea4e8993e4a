#!/bin/bash
# round 6, session 30: knn_collect_kernel -- the bound's rank over v_readlane in one wavefront, results as system-scope stores without the
# release fence: bit-exact tests, stamps, the closed loop from the compiled host
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s30; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_search.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
for q in 64 1; do echo "queries per call: $q"; GOCTR_DBG=knn goctr_amd/host/knn_bench --queries $q --steps 4 --warmup 20 --regions 1 2>&1 | grep knn_collect | tail -2; done | tee $O/knn_stamps.txt | cut -c1-420
for rep in 1 2; do for q in 1 64 256; do
echo "prev q=$q"; LD_LIBRARY_PATH=$R/goctr_amd/prevlib goctr_amd/host/knn_bench --queries $q --steps 200 --warmup 200 --regions 5
echo "new q=$q"; goctr_amd/host/knn_bench --queries $q --steps 200 --warmup 200 --regions 5
done; done 2>&1 | tee $O/knn_bench_q.txt | cut -c1-200
for rep in 1 2 3; do
timeout 300 python bench.py --workload knn --steps 200 --warmup 20 --no-cpu-baseline > $O/knn_$rep.json 2> $O/knn_$rep.err
python - <<P
import json
d=json.loads(open('$O/knn_$rep.json').read().strip().splitlines()[-1]); print('knn', d['value'], d['ms_per_step'], 'python loop', d['python_loop']['value'], '256/call', d['at_256_queries_per_call']['queries_per_s'])
P
done
