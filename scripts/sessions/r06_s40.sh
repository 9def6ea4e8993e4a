#!/bin/bash
# round 6, session 40: what a host synchronisation costs a short timed region (the driver's --steps 20): the runtime's default wait, its active
# wait (ROC_ACTIVE_WAIT_TIMEOUT), and hipDeviceScheduleSpin (GOCTR_EXP_SPIN, experiment knob)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s40; mkdir -p $O
for rep in 1 2; do
echo "== default"; timeout 300 python scripts/call_overhead.py 2>&1 | grep "K=   2\|K=  20\|K= 200"
echo "== ROC_ACTIVE_WAIT_TIMEOUT=2000"; ROC_ACTIVE_WAIT_TIMEOUT=2000 timeout 300 python scripts/call_overhead.py 2>&1 | grep "K=   2\|K=  20\|K= 200"
echo "== GOCTR_EXP_SPIN=1"; GOCTR_EXP_SPIN=1 timeout 300 python scripts/call_overhead.py 2>&1 | grep "K=   2\|K=  20\|K= 200"
done
drv() { n=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d['timed_regions_ms'])
P
}
for rep in 1 2 3; do
drv d_default_$rep
drv d_active_$rep ROC_ACTIVE_WAIT_TIMEOUT=2000
drv d_spin_$rep GOCTR_EXP_SPIN=1
done
