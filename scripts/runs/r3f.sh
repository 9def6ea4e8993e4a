#!/bin/bash
# round 3, GPU call F: item2vec with the corpus-length cap on the parallelism; all rows in the second tier (experiment)
O=gpurun_out/r3f; mkdir -p $O
GOCTR_W2V_T2=16384 timeout 300 python bench.py --workload item2vec --no-cpu-baseline > $O/i2v_t2_all.json 2> $O/i2v_t2_all.err
python - <<'P'
import json
try:
    d=json.loads(open('gpurun_out/r3f/i2v_t2_all.json').read().strip().splitlines()[-1]); print('T2=all', d['value'], d['ms_per_step'])
except Exception as e: print('ERR', e)
P
timeout 900 python -m pytest tests/test_gpu_w2v.py tests/test_gpu_corpus.py "tests/test_gpu_fullsize.py::test_item2vec_stress_point_v1e6_d64_hogwild_vs_oracle" -q -s -m gpu --timeout 800 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "HS loss|passed|failed|rc=|Error" $O/pytest.log | tail -10
GOCTR_W2V_T2=0 timeout 600 python -m pytest "tests/test_gpu_fullsize.py::test_item2vec_stress_point_v1e6_d64_hogwild_vs_oracle" -q -s -m gpu --timeout 500 -p no:cacheprovider > $O/pytest_t2off.log 2>&1
grep -E "HS loss|passed|failed" $O/pytest_t2off.log | tail -5
