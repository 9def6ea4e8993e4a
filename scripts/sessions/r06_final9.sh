#!/bin/bash
# round 6, closing session 9 (after the dependent-load fixes of sessions 60-65: every workload's kernels changed): rocprofv3 passes of all
# workloads, every bench line un-profiled, the whole -m gpu suite, smoke.  The merge back is capped at 64 MiB: the raw traces are pruned by
# prof_workload.sh.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_final9; mkdir -p $O
PREDICT=1 PASS_TIMEOUT=240 scripts/prof_workload.sh din > $O/prof_din.log 2>&1; tail -1 $O/prof_din.log
PREDICT=1 PASS_TIMEOUT=240 scripts/prof_workload.sh youtube --workload youtube > $O/prof_youtube.log 2>&1; tail -1 $O/prof_youtube.log
KT_EAGER=1 PASS_TIMEOUT=240 scripts/prof_workload.sh mlp --workload mlp > $O/prof_mlp.log 2>&1; tail -1 $O/prof_mlp.log
GOCTR_NO_GRAPH=1 PASSES=kt PASS_TIMEOUT=240 scripts/prof_workload.sh mlp100k --workload mlp100k --regions 1 > $O/prof_mlp100k.log 2>&1; tail -1 $O/prof_mlp100k.log
PASS_TIMEOUT=240 scripts/prof_workload.sh item2vec --workload item2vec > $O/prof_item2vec.log 2>&1; tail -1 $O/prof_item2vec.log
PASS_TIMEOUT=240 scripts/prof_workload.sh knn --workload knn > $O/prof_knn.log 2>&1; tail -1 $O/prof_knn.log
du -sh gpurun_out/p_*
scripts/bench_round.sh 2>&1 | tail -24
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
