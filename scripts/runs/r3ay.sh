#!/bin/bash
O=gpurun_out/r3ay; mkdir -p $O
for R in 1 2 3; do
for W in 5 20 36 100; do
timeout 300 python bench.py --no-cpu-baseline --no-serving --no-roofline --steps 20 --warmup $W > $O/x.json 2> $O/x.err
python - <<P
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('warmup=$W', d['value'], d['ms_per_step'])
P
done
done
