#!/bin/bash
O=gpurun_out/r3bl; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ctr.py tests/test_gpu_rank.py tests/test_gpu_assembly.py tests/test_gpu_fullsize.py tests/test_gpu_embtrain.py -q -m gpu -k "not item2vec" -p no:cacheprovider -x 2>&1 | tail -2
for R in 1 2; do
for W in "" "--workload youtube"; do
timeout 300 python bench.py $W --no-cpu-baseline --no-serving --steps 200 --warmup 20 > $O/x.json 2> $O/x.err
python - <<P
import json
d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('$W', d['value'], d['ms_per_step'], d.get('recommend_qps'), {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
done
done
