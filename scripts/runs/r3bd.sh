#!/bin/bash
O=gpurun_out/r3bd; mkdir -p $O
for M in 16 8 32 64 16; do
GOCTR_W2V_MERGE=$M timeout 300 python bench.py --workload item2vec --no-cpu-baseline > $O/x.json 2> $O/x.err
python - <<P
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('merge=$M', d['value'], d['ms_per_step'])
P
done
