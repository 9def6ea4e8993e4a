#!/bin/bash
# round 6, session 70: item2vec -- node updates deferred by TWO visits (libgoctr_hip_old.so = HEAD: one visit)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s70; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_w2v.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -1
for rep in 1 2 3; do
for L in libgoctr_hip_old.so libgoctr_hip.so; do
GOCTR_LIB=$R/goctr_amd/$L timeout 300 python bench.py --workload item2vec --no-cpu-baseline > $O/w_${L}_$rep.json 2>/dev/null
python3 -c "
import json; d=json.loads(open('$O/w_${L}_$rep.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'])"
done; done
