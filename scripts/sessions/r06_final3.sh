#!/bin/bash
# round 6, closing session 3 (after ctr_fwd4): DIN / YouTube rocprofv3 passes incl. predict, every bench line, the suite, smoke
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_final3; mkdir -p $O
PREDICT=1 PASS_TIMEOUT=300 scripts/prof_workload.sh din > $O/prof_din.log 2>&1
PREDICT=1 PASS_TIMEOUT=300 scripts/prof_workload.sh youtube --workload youtube > $O/prof_youtube.log 2>&1
scripts/bench_round.sh 2>&1 | tail -24
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
