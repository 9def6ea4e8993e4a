#!/bin/bash
O=gpurun_out/r3y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rank.py tests/test_gpu_ctr.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 30 $O/pytest.log
