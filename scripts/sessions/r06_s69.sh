#!/bin/bash
# round 6, session 69: k-NN collect -- one round of requests at the launch's start (query, ignore index, tile maxima) instead of two:
# tests, A/B against the previous library (libgoctr_hip_old.so), stamps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s69; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_search.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
run() {  # name, args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 400 python bench.py $wl --no-cpu-baseline > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], (d.get('timed_regions_ms') or [])[:5])
P
}
for rep in 1 2 3; do
run knn_old_$rep "--workload knn" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
run knn_new_$rep "--workload knn"
done
for L in libgoctr_hip_old.so libgoctr_hip.so; do
GOCTR_LIB=$R/goctr_amd/$L GOCTR_DBG=knn timeout 200 python bench.py --workload knn --steps 20 --warmup 5 --no-cpu-baseline --no-serving 2>&1 >/dev/null | grep -i "knn_collect\|sub-block" | tail -2
done
