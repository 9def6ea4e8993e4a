#!/bin/bash
O=gpurun_out/r3ah; mkdir -p $O
for R in 1 2 3; do
for L in libgoctr_hip.so libgoctr_hip_ra8.so; do
GOCTR_LIB=$PWD/goctr_amd/$L timeout 300 python bench.py --no-cpu-baseline --no-serving --no-roofline --phase train --steps 200 --warmup 20 > $O/din_${L}.json 2> $O/din_${L}.err
python - <<P
import json
d=json.loads(open('$O/din_${L}.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'])
P
done
done
