#!/bin/bash
O=gpurun_out/r3ae; mkdir -p $O
for R in 1 2; do
for L in libgoctr_hip.so libgoctr_hip_lb8.so; do
GOCTR_LIB=$PWD/goctr_amd/$L timeout 300 python bench.py --workload item2vec --no-cpu-baseline > $O/i2v_${L}.json 2> $O/i2v_${L}.err
python - <<P
import json
d=json.loads(open('$O/i2v_${L}.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'])
P
done
done
GOCTR_LIB=$PWD/goctr_amd/libgoctr_hip_lb8.so timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_w2v.py -q -m gpu -k "w2v or item2vec or hogwild" -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -12
