#!/bin/bash
# round 5, GPU session 5: the tests the last changes touch, k-NN kernel times, MLP phase stamps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s5; rm -rf $O; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_multi.py tests/test_gpu_ctr.py tests/test_gpu_comm.py -m gpu -q 2>&1 | tail -15) > $O/pytest.log
tail -6 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_knn -- python $R/bench.py --workload knn --no-cpu-baseline > $O/kt_knn.json 2> $O/kt_knn.err
f=$(ls $O/kt_knn/*/*_kernel_stats.csv | head -1); head -8 $f | cut -c1-200
find $O/kt_knn -type f ! -name '*_kernel_stats.csv' -delete
cd $R
timeout 200 python bench.py --workload knn --no-cpu-baseline > $O/knn.json 2> $O/knn.err
GOCTR_KNN_FOLD=0 timeout 200 python bench.py --workload knn --no-cpu-baseline > $O/knn_nofold.json 2> $O/knn_nofold.err
GOCTR_MLP_DBG=1 GOCTR_NO_GRAPH=1 timeout 200 python bench.py --workload mlp --no-cpu-baseline --steps 3 --warmup 0 > $O/mlp_dbg.json 2> $O/mlp_dbg.err
tail -8 $O/mlp_dbg.err
timeout 200 python bench.py --workload mlp --no-cpu-baseline > $O/mlp.json 2> $O/mlp.err
for f in $O/*.json; do python3 -c "
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['unit'], 'ms/step', d.get('ms_per_step'))
except Exception as e: print('$f', 'ERR', e)
"; done
