#!/bin/bash
# round 3, last GPU call: at HEAD -- the -m gpu suite, the rocprofv3 passes of the two CTR workloads (their attention kernels
# changed last), every bench line, the rank bench
O=gpurun_out/r3bm; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu --timeout 800 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 4 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
PREDICT=1 timeout 600 bash scripts/prof_workload.sh din > $O/p_din.log 2>&1
PREDICT=1 timeout 900 bash scripts/prof_workload.sh youtube --workload youtube > $O/p_youtube.log 2>&1
timeout 300 python bench.py > $O/bench_din.json 2> $O/bench_din.err
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_din_steps20.json 2> $O/bench_din_steps20.err
timeout 300 python bench.py --workload youtube > $O/bench_youtube.json 2> $O/bench_youtube.err
timeout 300 python bench.py --train-emb 0.05 > $O/bench_din_trainemb.json 2> $O/bench_din_trainemb.err
timeout 300 python bench.py --workload youtube --train-emb 0.05 > $O/bench_youtube_trainemb.json 2> $O/bench_youtube_trainemb.err
timeout 120 ./goctr_amd/host/rank_bench --threads 1,2,8 --n 32,256,2048 --seconds 0.5 > $O/rank_bench_din.json 2> $O/rank.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/rank_kt -- $GRAFT_REPO_ROOT/goctr_amd/host/rank_bench --threads 1 --n 256 --seconds 0.2 --coalesce 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/rank_kt.err
cd $GRAFT_REPO_ROOT
find $O/rank_kt -name '*_kernel_trace.csv' -delete
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3bm/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('recommend_qps'), d.get('recommend_qps_keys'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
P
