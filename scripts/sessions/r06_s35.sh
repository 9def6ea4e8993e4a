#!/bin/bash
# round 6, session 35: do two streams overlap one call's attention launch with another's forward chain?  (goctr_batch_predict from 1 / 2 / 4 threads)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for w in 2 1; do echo "GOCTR_FWD4_WGS=$w"; GOCTR_FWD4_WGS=$w timeout 300 python scripts/ubench/predict_two_streams.py 2>&1 | tail -6; done
echo "GOCTR_FWD4=0"; GOCTR_FWD4=0 timeout 300 python scripts/ubench/predict_two_streams.py 2>&1 | tail -6
