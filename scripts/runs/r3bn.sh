#!/bin/bash
O=gpurun_out/r3bn; mkdir -p $O
for F in 0 1; do
GOCTR_FORCE_COMM=$F timeout 300 python bench.py --no-cpu-baseline --no-serving --no-roofline --phase train --steps 200 --warmup 20 > $O/x.json 2> $O/x.err
python - <<P
import json
try:
    d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('force_comm=$F', d['value'], d['ms_per_step'], d.get('rccl_world'))
except Exception as e: print('ERR', e, open('$O/x.err').read()[-400:])
P
done
GOCTR_FORCE_COMM=1 timeout 300 python bench.py --workload youtube --no-cpu-baseline --no-serving --no-roofline --phase train --steps 200 --warmup 20 > $O/x.json 2> $O/x.err
python - <<P
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('youtube force_comm=1', d['value'], d['ms_per_step'])
P
