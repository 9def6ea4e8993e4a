#!/bin/bash
# every bench line of a round, un-profiled: scripts/gpu.sh -- scripts/bench_round.sh  ->  gpurun_out/bench/*.json (copy to profiles/rNN_bench_*.json)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/bench; rm -rf $O; mkdir -p $O
cd $R
timeout 400 python bench.py > $O/din.json 2> $O/din.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/din_steps20.json 2> $O/din_steps20.err
timeout 400 python bench.py --workload youtube > $O/youtube.json 2> $O/youtube.err
timeout 300 python bench.py --train-emb 0.05 > $O/din_trainemb.json 2> $O/din_trainemb.err
timeout 400 python bench.py --workload youtube --train-emb 0.05 > $O/youtube_trainemb.json 2> $O/youtube_trainemb.err
timeout 300 python bench.py --workload mlp > $O/mlp.json 2> $O/mlp.err
timeout 300 python bench.py --workload mlp100k > $O/mlp100k.json 2> $O/mlp100k.err
timeout 400 python bench.py --workload item2vec > $O/item2vec.json 2> $O/item2vec.err
timeout 300 python bench.py --workload knn > $O/knn.json 2> $O/knn.err
for f in $O/*.json; do python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['unit'], 'ms/step', d.get('ms_per_step'), 'qps', d.get('recommend_qps'), 'roofline', (d.get('roofline') or {}).get('frac'), 'ratio', d.get('step_traffic_ratio'))
except Exception as e: print('$f', 'ERR', e)
"; done
for f in $O/*.err; do tail -2 $f; done | head -40
