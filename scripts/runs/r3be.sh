#!/bin/bash
# round 3, GPU call O: at HEAD -- the whole -m gpu suite, the rocprofv3 passes of every workload (per phase), the bench lines
O=gpurun_out/r3be; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu --timeout 800 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 5 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
PREDICT=1 timeout 600 bash scripts/prof_workload.sh din > $O/p_din.log 2>&1
PREDICT=1 timeout 900 bash scripts/prof_workload.sh youtube --workload youtube > $O/p_youtube.log 2>&1
timeout 400 bash scripts/prof_workload.sh mlp --workload mlp > $O/p_mlp.log 2>&1
timeout 600 bash scripts/prof_workload.sh item2vec --workload item2vec > $O/p_item2vec.log 2>&1
timeout 400 bash scripts/prof_workload.sh knn --workload knn > $O/p_knn.log 2>&1
timeout 600 bash scripts/prof_workload.sh dinemb --train-emb 0.05 > $O/p_dinemb.log 2>&1
timeout 900 bash scripts/prof_workload.sh youtubeemb --workload youtube --train-emb 0.05 > $O/p_youtubeemb.log 2>&1
timeout 300 python bench.py > $O/bench_din.json 2> $O/bench_din.err
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_din_steps20.json 2> $O/bench_din_steps20.err
timeout 300 python bench.py --workload youtube > $O/bench_youtube.json 2> $O/bench_youtube.err
timeout 300 python bench.py --workload mlp > $O/bench_mlp.json 2> $O/bench_mlp.err
timeout 300 python bench.py --workload item2vec > $O/bench_item2vec.json 2> $O/bench_item2vec.err
timeout 300 python bench.py --workload knn > $O/bench_knn.json 2> $O/bench_knn.err
timeout 300 python bench.py --train-emb 0.05 > $O/bench_din_trainemb.json 2> $O/bench_din_trainemb.err
timeout 300 python bench.py --workload youtube --train-emb 0.05 > $O/bench_youtube_trainemb.json 2> $O/bench_youtube_trainemb.err
timeout 120 ./goctr_amd/host/rank_bench --threads 1,2,8 --n 32,256,2048 --seconds 0.5 > $O/rank_bench_din.json 2> $O/rank.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/rank_kt -- $GRAFT_REPO_ROOT/goctr_amd/host/rank_bench --threads 1 --n 256 --seconds 0.2 --coalesce 1 > /dev/null 2> $GRAFT_REPO_ROOT/$O/rank_kt.err
cd $GRAFT_REPO_ROOT
find $O/rank_kt -name '*_kernel_trace.csv' -delete
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3be/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
P
