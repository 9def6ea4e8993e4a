#!/bin/bash
# round 6: every bench line un-profiled on the final tree (scripts/bench_round.sh) + the -m gpu suite + smoke
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_bench; mkdir -p $O
scripts/bench_round.sh 2>&1 | tail -24
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
