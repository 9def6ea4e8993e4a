#!/bin/bash
# round 5, session 16: slab heights of the weight-gradient launch around the searched optimum (GOCTR_TN_C0 / C1 = 32-row chunks per
# dW0 / dW1 workgroup, GOCTR_TN_RL = rows per one-tile workgroup): the whole cfg3 step, median of 5 regions of 200 steps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s16; rm -rf $O; mkdir -p $O
cd $R
run() { # name c0 c1 rl
  GOCTR_TN_C0=$2 GOCTR_TN_C1=$3 GOCTR_TN_RL=$4 timeout 200 python bench.py --steps 200 --warmup 5 --regions 5 --no-cpu-baseline --no-serving --no-roofline 2>/dev/null | tail -1 > $O/$1.json
  python3 -c "
import json; d=json.loads(open('$O/$1.json').read()); print('$1 c0=$2 c1=$3 rl=$4', d['value'], d['ms_per_step'], d['timed_regions_ms'])"
}
for rep in 1 2; do
run default_$rep 0 0 0
run half_chip_$rep 8 12 784
run c3_5_$rep 3 5 392
run c5_6_328_$rep 5 6 328
run c4_6_256_$rep 4 6 256
run c4_5_512_$rep 4 5 512
run c6_8_512_$rep 6 8 512
done | tee $O/summary.txt
