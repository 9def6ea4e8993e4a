#!/bin/bash
O=gpurun_out/r3az; mkdir -p $O
for R in 1 2; do
for ROWS in 262144 1048576 4194304; do
timeout 300 python bench.py --no-cpu-baseline --no-serving --no-roofline --phase train --steps 200 --warmup 20 --rows $ROWS > $O/x.json 2> $O/x.err
python - <<P
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('rows=$ROWS', d['value'], d['ms_per_step'])
P
done
done
