#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
for v in "sum:GOCTR_W2V_T2_SUM=1" "off:GOCTR_W2V_T2=0"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 600 python -m pytest "tests/test_gpu_fullsize.py::test_cfg5_item2vec_full_size_hogwild_vs_oracle" "tests/test_gpu_fullsize.py::test_item2vec_stress_point_v1e6_d64_hogwild_vs_oracle" -q -s -m gpu --timeout 500 -p no:cacheprovider > $O/pytest_$n.log 2>&1
  echo "== $n"; grep -E "HS loss|same-topic|passed|failed" $O/pytest_$n.log | tail -6
done
