#!/bin/bash
# round 6, session 17: knn_collect_kernel -- the bound from per-wavefront top lists, the tile list from registers behind one barrier, tie-free
# insertions as a DPP shift: bit-exact tests, stamps, the line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_search.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -3
for q in 64 1; do echo "queries per call: $q"; GOCTR_DBG=knn goctr_amd/host/knn_bench --queries $q --steps 4 --warmup 20 --regions 1 2>&1 | grep knn_collect | tail -2; done | tee $O/knn_stamps.txt | cut -c1-420
for q in 1 8 64 256 1024; do goctr_amd/host/knn_bench --queries $q --steps 200 --warmup 200 --regions 5; done 2>&1 | tee $O/knn_bench_q.txt | cut -c1-200
timeout 300 python bench.py --workload knn --steps 200 --warmup 20 --no-cpu-baseline > $O/knn.json 2> $O/knn.err
python - <<P
import json
d=json.loads(open('$O/knn.json').read().strip().splitlines()[-1]); print('knn', d['value'], d['ms_per_step'], d['timed_regions_ms'], 'python loop', d['python_loop']['value'])
P
