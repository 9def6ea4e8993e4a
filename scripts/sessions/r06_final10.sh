#!/bin/bash
# round 6, closing session 10 (HEAD after session 66's chain change): rocprofv3 passes of din / youtube incl. predict, the DIN lines,
# the whole -m gpu suite, smoke
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_final10; mkdir -p $O
PREDICT=1 PASS_TIMEOUT=240 scripts/prof_workload.sh din > $O/prof_din.log 2>&1; tail -1 $O/prof_din.log
PREDICT=1 PASS_TIMEOUT=240 scripts/prof_workload.sh youtube --workload youtube > $O/prof_youtube.log 2>&1; tail -1 $O/prof_youtube.log
scripts/bench_round.sh 2>&1 | tail -24
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
