#!/bin/bash
O=gpurun_out/r3ap; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ctr.py tests/test_gpu_fullsize.py tests/test_gpu_rank.py tests/test_gpu_assembly.py -q -m gpu -k "not item2vec" -p no:cacheprovider 2>&1 | tail -5
for W in "" "--workload youtube"; do
for G in 8 16; do
GOCTR_PRED_GROUP=$G timeout 300 python bench.py --no-cpu-baseline --no-serving --steps 200 --warmup 20 $W > $O/x.json 2> $O/x.err
python - <<PY
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('$W group=$G', d['value'], d.get('recommend_qps'), d.get('recommend_qps_keys'))
PY
done
done
