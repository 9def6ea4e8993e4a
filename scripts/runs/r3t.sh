#!/bin/bash
O=gpurun_out/r3t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ctr.py tests/test_gpu_fullsize.py::test_cfg3_din_full_size_step_vs_oracle tests/test_gpu_comm.py -q -m gpu --timeout 600 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 8 $O/pytest.log
for C in 1 0 1 0; do
GOCTR_CHAIN_ATTN_BWD=$C timeout 300 python bench.py --no-cpu-baseline --no-serving --steps 200 --warmup 20 > $O/din_ab$C.json 2> $O/din_ab$C.err
python - <<P
import json
d=json.loads(open('$O/din_ab$C.json').read().strip().splitlines()[-1]); print($C, d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
done
GOCTR_CHAIN_DBG=1 timeout 120 python bench.py --no-cpu-baseline --no-serving --steps 3 --warmup 1 --phase train 2>&1 | grep "chain_x3 phases" | tail -2
GOCTR_CHAIN_ATTN_BWD=0 GOCTR_CHAIN_DBG=1 timeout 120 python bench.py --no-cpu-baseline --no-serving --steps 3 --warmup 1 --phase train 2>&1 | grep "chain_x3 phases" | tail -2
