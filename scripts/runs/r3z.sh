#!/bin/bash
O=gpurun_out/r3z; mkdir -p $O
for L in libgoctr_hip.so libgoctr_hip_s2.so libgoctr_hip_s3.so libgoctr_hip_s4.so libgoctr_hip.so libgoctr_hip_s2.so; do
GOCTR_LIB=$PWD/goctr_amd/$L timeout 300 python bench.py --no-cpu-baseline --no-serving --steps 200 --warmup 20 > $O/din_${L}.json 2> $O/din_${L}.err
python - <<P
import json
d=json.loads(open('$O/din_${L}.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
done
for L in libgoctr_hip_s2.so libgoctr_hip_s4.so; do
GOCTR_LIB=$PWD/goctr_amd/$L timeout 120 python scripts/dbg_chain.py 2>&1 | grep "chain_x3 phases" | tail -1
GOCTR_LIB=$PWD/goctr_amd/$L timeout 300 python -m pytest tests/test_gpu_ctr.py -q -m gpu -k "attention_backward or pipelined" -p no:cacheprovider 2>&1 | tail -1
done
