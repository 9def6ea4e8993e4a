#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_comm.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -4
