#!/bin/bash
# round 6, session 68: item2vec -- a visit's node update issued at the head of the next visit (libgoctr_hip_old.so = HEAD's library)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s68; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_w2v.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multi.py -q -m gpu -x -p no:cacheprovider -k "w2v or item2vec or hogwild or embedding" 2>&1 | tail -2
run() {  # name, args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 400 python bench.py $wl --no-cpu-baseline > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d.get('hs_loss') or d.get('loss') or '', (d.get('timed_regions_ms') or [])[:5])
P
}
for rep in 1 2 3; do
run w2v_old_$rep "--workload item2vec" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
run w2v_new_$rep "--workload item2vec"
done
