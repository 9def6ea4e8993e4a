#!/bin/bash
# round 5, closing session: full -m gpu suite, smoke, every bench line un-profiled (the kernels timed under rocprofv3 in r05_final.sh
# are unchanged; the serving pass and the k-NN call gained their completion words since)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final2; rm -rf $O; mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/pytest.log; tail -4 $O/pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $O/smoke.log; cat $O/smoke.log
scripts/bench_round.sh > $O/bench_round.log 2>&1; tail -30 $O/bench_round.log
