#!/bin/bash
# round 3, GPU call E: item2vec second tier (per-XCD L2 replicas) A/B + statistical gates
O=gpurun_out/r3e; mkdir -p $O
for cfgname in "t2_1024:GOCTR_W2V_T2=1024" "t2_off:GOCTR_W2V_T2=0" "t2_1024_sum:GOCTR_W2V_T2=1024 GOCTR_W2V_T2_SUM=1" "t2_4096:GOCTR_W2V_T2=4096"; do
  name=${cfgname%%:*}; envs=${cfgname#*:}
  env $envs timeout 300 python bench.py --workload item2vec --no-cpu-baseline > $O/i2v_$name.json 2> $O/i2v_$name.err
  python - "$O/i2v_$name.json" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'])
except Exception as e: print(sys.argv[1], 'ERR', e)
P
done
timeout 900 python -m pytest tests/test_gpu_w2v.py "tests/test_gpu_fullsize.py::test_cfg5_item2vec_full_size_hogwild_vs_oracle" "tests/test_gpu_fullsize.py::test_item2vec_stress_point_v1e6_d64_hogwild_vs_oracle" -q -s -m gpu --timeout 800 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "HS loss|same-topic|passed|failed|rc=|Error|error" $O/pytest.log | tail -20
GOCTR_W2V_T2=0 timeout 600 python -m pytest "tests/test_gpu_fullsize.py::test_cfg5_item2vec_full_size_hogwild_vs_oracle" -q -s -m gpu --timeout 500 -p no:cacheprovider > $O/pytest_t2off.log 2>&1
grep -E "HS loss|same-topic|passed|failed" $O/pytest_t2off.log | tail -5
PASSES="kt fetch write" timeout 400 bash scripts/prof_workload.sh item2vec --workload item2vec > /dev/null 2>&1
python scripts/prof_summarize.py r03_item2vec gpurun_out/p_item2vec | head -12
