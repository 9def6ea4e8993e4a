#!/bin/bash
# round 6, session 8: the whole -m gpu suite on the tree with XCD-affine dealing, slabs through the L2, per-tile sums; smoke; k-NN scan
# maxima through the L2 (A/B, 4 repetitions); rocprofv3 kernel trace of the mlp100k fit
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s8; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
for rep in 1 2 3 4; do for x in 0 1; do
GOCTR_KNN_WT=$x timeout 300 python bench.py --workload knn --steps 200 --warmup 20 --no-cpu-baseline > $O/knn_wt${x}_$rep.json 2> $O/knn_wt${x}_$rep.err
python - <<P
import json
d=json.loads(open('$O/knn_wt${x}_$rep.json').read().strip().splitlines()[-1]); print('knn wt=$x rep $rep', d['value'], d['ms_per_step'], d['timed_regions_ms'])
P
done; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_mlp100k -o mlp100k -- python $R/bench.py --workload mlp100k --no-cpu-baseline --regions 2 > $R/$O/mlp100k_prof.json 2> $R/$O/mlp100k_prof.err
cd $R; ls $O/prof_mlp100k | head; find $O/prof_mlp100k -name "*kernel_stats*" | head -2 | while read f; do head -12 "$f"; done
