#!/bin/bash
# round 6, closing session 13 (HEAD after session 69's k-NN collect change): kernel trace of knn, the whole -m gpu suite, smoke
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_final13; mkdir -p $O gpurun_out/bench
PASS_TIMEOUT=240 scripts/prof_workload.sh knn --workload knn > $O/prof_knn.log 2>&1; tail -1 $O/prof_knn.log
timeout 300 python bench.py --workload knn > gpurun_out/bench/knn.json 2> gpurun_out/bench/knn.err; python3 -c "
import json; d=json.loads(open('gpurun_out/bench/knn.json').read().strip().splitlines()[-1]); print('knn', d['value'], d['ms_per_step'])"
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
