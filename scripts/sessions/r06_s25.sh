#!/bin/bash
# round 6, session 25: per-workgroup start / end / placement of ctr_fwd4 (GOCTR_FWD4_DUMP)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s25; mkdir -p $O
for v in 1 2; do
  GOCTR_FWD4_DUMP=$O/wgs$v.txt GOCTR_FWD4_WGS=$v GOCTR_DBG=chain timeout 300 python scripts/ubench/fwd_phases.py 2>&1 | grep "fwd4" | tail -1
done
