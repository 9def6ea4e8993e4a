#!/bin/bash
timeout 120 python scripts/dbg_predict.py 2>&1 | grep "forward-only" | tail -4
