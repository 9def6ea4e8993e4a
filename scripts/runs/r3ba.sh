#!/bin/bash
O=gpurun_out/r3ba; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ctr.py tests/test_gpu_comm.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -2
for R in 1 2 3; do
for P in 1 0; do
GOCTR_DATASET_PREFETCH=$P timeout 300 python bench.py --no-cpu-baseline --no-serving --no-roofline --phase train --steps 200 --warmup 20 --rows 1048576 > $O/x.json 2> $O/x.err
python - <<PY
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('rows=2^20 prefetch=$P', d['value'], d['ms_per_step'])
PY
GOCTR_DATASET_PREFETCH=$P timeout 300 python bench.py --no-cpu-baseline --no-serving --no-roofline --steps 20 --warmup 5 > $O/x.json 2> $O/x.err
python - <<PY
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('K=20 W=5 prefetch=$P', d['value'], d['ms_per_step'])
PY
done
done
for P in 1 0; do
GOCTR_DATASET_PREFETCH=$P timeout 300 python bench.py --workload youtube --no-cpu-baseline --no-serving --no-roofline --steps 20 --warmup 5 > $O/x.json 2> $O/x.err
python - <<PY
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('youtube K=20 W=5 prefetch=$P', d['value'], d['ms_per_step'])
PY
done
