#!/bin/bash
O=gpurun_out/r3bc; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ctr.py tests/test_gpu_embtrain.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
for R in 1 2 3; do
for P in 1 0; do
GOCTR_H0_CARRY=$P timeout 300 python bench.py --no-cpu-baseline --no-serving --no-roofline --steps 20 --warmup 5 > $O/x.json 2> $O/x.err
python - <<PY
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('K=20 W=5 carry=$P', d['value'], d['ms_per_step'])
PY
done
done
