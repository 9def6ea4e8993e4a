#!/bin/bash
# round 6, session 46: HEAD (att0 inside the weight-gradient launch; the last launch's scratch use of session 43-45's builds fixed) against HEAD
# with the reduce block + flag (GOCTR_ATT0_EARLY=0) and against the library of commit b293275 (libgoctr_hip_old.so), one box; then the counters
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s46; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -2
train() {  # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'])
P
}
for rep in 1 2 3 4; do
train head_$rep
train flag_$rep GOCTR_ATT0_EARLY=0
train old_$rep GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
done
bash scripts/sessions/r06_s45.sh 2>&1 | grep "reduce_attn\|x3w"
