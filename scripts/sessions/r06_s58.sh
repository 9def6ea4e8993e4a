#!/bin/bash
# round 6, session 58: where the kernel arguments live -- HIP_FORCE_DEV_KERNARG=0 / 1 / unset (the runtime's default) on the DIN line,
# the driver's 20-step form, the MLP line and the k-NN line; interleaved
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s58; mkdir -p $O
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --no-cpu-baseline --no-serving > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], (d.get('timed_regions_ms') or [])[1:5])
P
}
for rep in 1 2; do
run din_unset_$rep "--steps 200 --warmup 20" A=1
run din_k0_$rep "--steps 200 --warmup 20" HIP_FORCE_DEV_KERNARG=0
run din_k1_$rep "--steps 200 --warmup 20" HIP_FORCE_DEV_KERNARG=1
done
run din20_k0 "--steps 20 --warmup 5" HIP_FORCE_DEV_KERNARG=0
run din20_k1 "--steps 20 --warmup 5" HIP_FORCE_DEV_KERNARG=1
run mlp_k0 "--workload mlp --steps 200 --warmup 20" HIP_FORCE_DEV_KERNARG=0
run mlp_k1 "--workload mlp --steps 200 --warmup 20" HIP_FORCE_DEV_KERNARG=1
