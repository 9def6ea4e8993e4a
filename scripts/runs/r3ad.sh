#!/bin/bash
O=gpurun_out/r3ad; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ctr.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py::test_cfg3_din_full_size_step_vs_oracle -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 3 $O/pytest.log
for W in "" "--workload youtube"; do
timeout 300 python bench.py --no-cpu-baseline --no-serving --steps 200 --warmup 20 $W > $O/b.json 2> $O/b.err
python - <<P
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('$W', d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()})
P
done
PREDICT=1 timeout 600 bash scripts/prof_workload.sh din > $O/p_din.log 2>&1
