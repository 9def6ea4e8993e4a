#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_comm.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
GOCTR_FORCE_COMM=1 timeout 200 python scripts/dp_step_time.py 2>/dev/null | grep "comm="
GOCTR_FORCE_COMM=1 GOCTR_DP_JOIN_GRAPHS=0 timeout 200 python scripts/dp_step_time.py 2>/dev/null | grep "comm="
