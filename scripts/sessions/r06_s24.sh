#!/bin/bash
# round 6, session 24: shader clock during the forward-only kernels (s_memtime against s_memrealtime), 1 / 2 workgroups per CU
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for v in 1 2 3; do
  echo "GOCTR_FWD4_WGS=$v"; GOCTR_FWD4_WGS=$v GOCTR_DBG=chain timeout 300 python scripts/ubench/fwd_phases.py 2>&1 | grep "fwd4" | tail -2
done
echo "GOCTR_FWD4=0"; GOCTR_FWD4=0 GOCTR_DBG=chain timeout 300 python scripts/ubench/fwd_phases.py 2>&1 | grep "chain_x3" | tail -2
