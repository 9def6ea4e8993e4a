#!/bin/bash
O=gpurun_out/r3p; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rank.py tests/test_gpu_assembly.py -q -m gpu --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 4 $O/pytest.log
timeout 120 ./goctr_amd/host/rank_bench --threads 1,2,8 --n 32,256,2048 --seconds 0.5 > $O/rank_bench_din.json 2> $O/rank.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/rank_kt -- $GRAFT_REPO_ROOT/goctr_amd/host/rank_bench --threads 1 --n 256 --seconds 0.2 --coalesce 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/rank_kt.err
cd $GRAFT_REPO_ROOT
find $O/rank_kt -name '*_kernel_trace.csv' -delete
python - <<'P'
import json,csv,glob
d=json.load(open('gpurun_out/r3p/rank_bench_din.json'))
for e in d['results']: print(e['n'], e['threads'], e['coalesce'], round(e['rank_qps']), e['latency_us']['p50'], e['latency_us']['p99'])
for r in csv.DictReader(open(glob.glob('gpurun_out/r3p/rank_kt/**/*_kernel_stats.csv', recursive=True)[0])): print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2))
P
