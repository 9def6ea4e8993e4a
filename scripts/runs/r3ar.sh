#!/bin/bash
O=gpurun_out/r3ar; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rank.py tests/test_gpu_assembly.py tests/test_gpu_ctr.py -q -m gpu --timeout 300 -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
for R in 1 2; do
for F in 1 0; do
GOCTR_SERVE_ONE_LAUNCH=$F timeout 120 ./goctr_amd/host/rank_bench --threads 1,8 --n 32,256,2048 --seconds 0.4 --coalesce 1 > $O/rank_f$F.json 2> $O/rank_f$F.err
python - <<P
import json
try:
    d=json.load(open('gpurun_out/r3ar/rank_f$F.json'))
    print('one_launch=$F', [(r['n'], r['threads'], round(r['rank_qps']), r['latency_us']['p50'], r['mismatched_calls']) for r in d['results']])
except Exception as e: print('$F','ERR',e, open('gpurun_out/r3ar/rank_f$F.err').read()[-500:])
P
done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/rank_kt -- $GRAFT_REPO_ROOT/goctr_amd/host/rank_bench --threads 1 --n 256 --seconds 0.2 --coalesce 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/rank_kt.err
head -4 $(ls $GRAFT_REPO_ROOT/$O/rank_kt/*/*_kernel_stats.csv | head -1) | cut -c1-150
