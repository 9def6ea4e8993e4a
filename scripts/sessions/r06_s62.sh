#!/bin/bash
# round 6, session 62: + the chain tail's four factor loads in one round (libgoctr_hip_old.so = session 61's library)
# behind its sample's short-circuit tests), the step counter with the first state load, and the weight-gradient launch's operand
# descriptors pinned as scalars (they were a per-lane load from the argument buffer in front of chunk 0): tests, A/B against HEAD's
# library (libgoctr_hip_old.so) on one box, stamps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s62; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_pipeline.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
run() {  # name, args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], (d.get('timed_regions_ms') or [])[1:6])
P
}
for rep in 1 2 3; do
run din_old_$rep "--steps 200 --warmup 20" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
run din_new_$rep "--steps 200 --warmup 20"
done
run din20_old "--steps 20 --warmup 5" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
run din20_new "--steps 20 --warmup 5"
run yt_old "--workload youtube --steps 200 --warmup 20" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
run yt_new "--workload youtube --steps 200 --warmup 20"
for L in libgoctr_hip_old.so libgoctr_hip.so; do
GOCTR_LIB=$R/goctr_amd/$L GOCTR_DBG=chain,tn GOCTR_NO_GRAPH=1 timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-serving --no-roofline 2>&1 >/dev/null | grep -i "chain\|tn\|wait" | tail -6
done
