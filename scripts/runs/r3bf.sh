#!/bin/bash
O=gpurun_out/r3bf; mkdir -p $O
for R in 1 2 3; do
for L in libgoctr_hip.so libgoctr_hip_r4.so; do
GOCTR_LIB=$PWD/goctr_amd/$L timeout 300 python bench.py --workload youtube --no-cpu-baseline --no-serving --no-roofline --phase predict --steps 200 --warmup 20 > $O/x.json 2> $O/x.err
python - <<P
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('$L', d.get('recommend_qps'))
P
done
done
