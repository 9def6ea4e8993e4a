#!/bin/bash
# round 6, closing session 11 (HEAD after session 68's item2vec change): rocprofv3 passes of item2vec, every bench line, the whole -m gpu
# suite, smoke
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_final11; mkdir -p $O
PASS_TIMEOUT=240 scripts/prof_workload.sh item2vec --workload item2vec > $O/prof_item2vec.log 2>&1; tail -1 $O/prof_item2vec.log
scripts/bench_round.sh 2>&1 | tail -24
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
