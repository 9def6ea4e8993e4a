#!/bin/bash
GOCTR_FORCE_COMM=1 timeout 200 python scripts/dp_step_time.py 2>/dev/null | grep "comm="
