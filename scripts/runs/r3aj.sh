#!/bin/bash
O=gpurun_out/r3aj; mkdir -p $O
for R in 1 2 3; do
for L in libgoctr_hip.so libgoctr_hip_old.so; do
GOCTR_LIB=$PWD/goctr_amd/$L timeout 300 python bench.py --no-cpu-baseline --no-serving --steps 200 --warmup 20 > $O/din_${L}.json 2> $O/din_${L}.err
python - <<P
import json
d=json.loads(open('$O/din_${L}.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], d.get('recommend_qps'), {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
done
done
for L in libgoctr_hip.so libgoctr_hip_old.so; do
GOCTR_LIB=$PWD/goctr_amd/$L timeout 300 python bench.py --no-cpu-baseline --no-serving --train-emb 0.05 --steps 200 --warmup 20 > $O/dinemb_${L}.json 2> $O/dinemb_${L}.err
python - <<P
import json
d=json.loads(open('$O/dinemb_${L}.json').read().strip().splitlines()[-1]); print('emb $L', d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
done
timeout 600 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_rank.py tests/test_gpu_embtrain.py tests/test_gpu_pipeline.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
