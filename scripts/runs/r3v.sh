#!/bin/bash
O=gpurun_out/r3v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ctr.py -q -m gpu --timeout 600 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 4 $O/pytest.log
for L in libgoctr_hip.so libgoctr_hip_ab0.so; do
for C in 1 0; do
GOCTR_LIB=$PWD/goctr_amd/$L GOCTR_CHAIN_ATTN_BWD=$C timeout 300 python bench.py --no-cpu-baseline --no-serving --steps 200 --warmup 20 > $O/din_${L}_$C.json 2> $O/din_${L}_$C.err
python - <<P
import json
d=json.loads(open('$O/din_${L}_$C.json').read().strip().splitlines()[-1]); print('$L', $C, d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
done
GOCTR_LIB=$PWD/goctr_amd/$L timeout 120 python scripts/dbg_chain.py 2>&1 | grep "chain_x3 phases" | tail -1
done
