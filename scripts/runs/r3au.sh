#!/bin/bash
O=gpurun_out/r3au; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rank.py tests/test_gpu_assembly.py -q -m gpu --timeout 300 -p no:cacheprovider -x 2>&1 | tail -2
for R in 1 2; do
timeout 120 ./goctr_amd/host/rank_bench --threads 1,8 --n 32,256,2048 --seconds 0.4 --coalesce 1 > $O/rank.json 2> $O/rank.err
python - <<P
import json
d=json.load(open('gpurun_out/r3au/rank.json'))
print([(r['n'], r['threads'], round(r['rank_qps']), r['latency_us']['p50'], r['mismatched_calls']) for r in d['results']])
P
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -- $GRAFT_REPO_ROOT/goctr_amd/host/rank_bench --threads 1 --n 256 --seconds 0.2 --coalesce 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/kt.err
head -2 $(ls $GRAFT_REPO_ROOT/$O/kt/*/*_kernel_stats.csv | head -1) | cut -c1-150
