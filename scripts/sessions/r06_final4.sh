#!/bin/bash
# round 6, closing session 4 (ctr_fwd4 at both shapes, k-NN results without the fence): rocprofv3 passes of din / youtube (incl. predict) / knn,
# every bench line, the whole -m gpu suite twice, smoke
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_final4; mkdir -p $O
PASS_TIMEOUT=300 scripts/prof_workload.sh knn --workload knn > $O/prof_knn.log 2>&1
PREDICT=1 PASS_TIMEOUT=300 scripts/prof_workload.sh din > $O/prof_din.log 2>&1
PREDICT=1 PASS_TIMEOUT=300 scripts/prof_workload.sh youtube --workload youtube > $O/prof_youtube.log 2>&1
scripts/bench_round.sh 2>&1 | tail -24
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest2.log 2>&1; tail -4 $O/pytest2.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
