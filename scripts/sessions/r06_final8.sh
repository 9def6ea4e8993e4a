#!/bin/bash
# round 6, closing session 8: rocprofv3 passes of the sklearn-port MLP workloads (float64 row image + prefetch blocks), every bench line
# un-profiled, the whole -m gpu suite, smoke
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_final8; mkdir -p $O
KT_EAGER=1 PASS_TIMEOUT=240 scripts/prof_workload.sh mlp --workload mlp > $O/prof_mlp.log 2>&1; tail -1 $O/prof_mlp.log
GOCTR_NO_GRAPH=1 PASSES=kt PASS_TIMEOUT=240 scripts/prof_workload.sh mlp100k --workload mlp100k --regions 1 > $O/prof_mlp100k.log 2>&1; tail -1 $O/prof_mlp100k.log
scripts/bench_round.sh 2>&1 | tail -24
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
