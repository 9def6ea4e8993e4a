#!/bin/bash
O=gpurun_out/r3x; mkdir -p $O
timeout 300 python scripts/call_overhead.py 2>&1 | tee $O/call_overhead.txt
