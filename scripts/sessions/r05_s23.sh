#!/bin/bash
# round 5, session 23: collect kernel scores the listed sub-blocks' rows exactly right away (no float32 pass in front) when there are
# at most 512 of them: search tests, bench line, kernel durations
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s23; rm -rf $O; mkdir -p $O
cd $R
(timeout 500 python -m pytest tests/test_gpu_search.py -m gpu -q 2>&1 | tail -6) > $O/pytest.log; tail -3 $O/pytest.log
for rep in 1 2 3; do timeout 200 python bench.py --workload knn --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['timed_regions_ms'])"; done | tee $O/bench.txt
cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kk -- python $R/bench.py --workload knn --steps 100 --regions 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1; f=$(find /tmp/kk -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | cut -c1-120 | head -5 | tee $O/kernels.txt
