#!/bin/bash
# round 5, GPU session 2: full -m gpu suite on the forked pipeline + folded k-NN, then A/B lines
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s2; rm -rf $O; mkdir -p $O
cd $R
(timeout 1300 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -80) > $O/pytest.log
tail -4 $O/pytest.log
B="--no-cpu-baseline --no-serving --no-roofline"
for f in 1 0; do
  GOCTR_FORK_ATTN=$f timeout 200 python bench.py $B > $O/din_fork$f.json 2> $O/din_fork$f.err
  GOCTR_FORK_ATTN=$f timeout 200 python bench.py $B --steps 20 --warmup 5 > $O/din20_fork$f.json 2> $O/din20_fork$f.err
  GOCTR_FORK_ATTN=$f timeout 300 python bench.py $B --workload youtube > $O/yt_fork$f.json 2> $O/yt_fork$f.err
done
GOCTR_FORK_ATTN=0 GOCTR_ATT0_EARLY=0 timeout 200 python bench.py $B > $O/din_r4path.json 2> $O/din_r4path.err
for g in 8 16 32; do
  GOCTR_KNN_G=$g timeout 200 python bench.py --workload knn --no-cpu-baseline > $O/knn_g$g.json 2> $O/knn_g$g.err
done
GOCTR_KNN_FOLD=0 timeout 200 python bench.py --workload knn --no-cpu-baseline > $O/knn_nofold.json 2> $O/knn_nofold.err
timeout 300 python bench.py --workload item2vec --no-cpu-baseline > $O/item2vec.json 2> $O/item2vec.err
for f in $O/*.json; do python3 -c "
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['unit'], 'ms/step', d.get('ms_per_step'), 'qps', d.get('recommend_qps'), 'spread', d.get('timed_region_spread'))
except Exception as e: print('$f', 'ERR', e)
"; done
for f in $O/*.err; do echo "== $f"; tail -3 $f; done 2>/dev/null | head -60
