#!/bin/bash
O=gpurun_out/r3aa; mkdir -p $O
for L in libgoctr_hip.so libgoctr_hip_tb.so libgoctr_hip.so libgoctr_hip_tb.so; do
GOCTR_LIB=$PWD/goctr_amd/$L timeout 300 python bench.py --no-cpu-baseline --no-serving --steps 200 --warmup 20 > $O/din_${L}.json 2> $O/din_${L}.err
python - <<P
import json
d=json.loads(open('$O/din_${L}.json').read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
done
for L in libgoctr_hip.so libgoctr_hip_tb.so; do
GOCTR_LIB=$PWD/goctr_amd/$L timeout 120 python scripts/dbg_chain.py 2>&1 | grep "chain_x3 phases" | tail -1
GOCTR_LIB=$PWD/goctr_amd/$L timeout 300 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_pipeline.py -q -m gpu -p no:cacheprovider 2>&1 | tail -1
done
