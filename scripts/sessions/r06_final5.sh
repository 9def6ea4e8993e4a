#!/bin/bash
# round 6, closing session 5, part 1 (the attention forward's one factor instead of gate + weight): rocprofv3 passes of din / youtube incl. predict
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_final5; mkdir -p $O
PREDICT=1 PASS_TIMEOUT=300 scripts/prof_workload.sh din > $O/prof_din.log 2>&1; tail -3 $O/prof_din.log
PREDICT=1 PASS_TIMEOUT=300 scripts/prof_workload.sh youtube --workload youtube > $O/prof_youtube.log 2>&1; tail -3 $O/prof_youtube.log
