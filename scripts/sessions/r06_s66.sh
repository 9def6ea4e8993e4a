#!/bin/bash
# round 6, session 66: the chain launch's slot ids tested behind the h0 split's barrier instead of in the prologue (libgoctr_hip_old.so =
# closing session 9's library): tests, A/B, stamps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s66; mkdir -p $O
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
for L in libgoctr_hip_old.so libgoctr_hip.so; do
GOCTR_LIB=$R/goctr_amd/$L GOCTR_DBG=chain GOCTR_NO_GRAPH=1 timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-serving --no-roofline 2>&1 >/dev/null | grep -i "chain_x3 phases" | tail -3
done
