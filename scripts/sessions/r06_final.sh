#!/bin/bash
# round 6, closing session: rocprofv3 passes of the workloads whose kernels changed since r06_prof1-3 (knn, mlp, din / youtube attention),
# every bench line un-profiled, the whole -m gpu suite, smoke
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_final; mkdir -p $O
PASS_TIMEOUT=300 scripts/prof_workload.sh knn --workload knn > $O/prof_knn.log 2>&1
KT_EAGER=1 PASS_TIMEOUT=300 scripts/prof_workload.sh mlp --workload mlp > $O/prof_mlp.log 2>&1
PREDICT=1 PASS_TIMEOUT=300 scripts/prof_workload.sh din > $O/prof_din.log 2>&1
PREDICT=1 PASS_TIMEOUT=300 scripts/prof_workload.sh youtube --workload youtube > $O/prof_youtube.log 2>&1
du -sh gpurun_out/p_knn gpurun_out/p_mlp gpurun_out/p_din gpurun_out/p_youtube
scripts/bench_round.sh 2>&1 | tail -24
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
