#!/bin/bash
# round 6, session 59: MLP weight-gradient slabs in the blocked layout [256-parameter block][slab][256] (GOCTR_MLP_SLAB_BLOCKED): tests, A/B, stamps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s59; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_fullsize.py -q -m gpu -x -p no:cacheprovider -k "mlp or Mlp or flagship or sklearn" 2>&1 | tail -2
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d.get('us_per_update'), d.get('timed_regions_ms')[1:6])
P
}
for rep in 1 2 3; do
for m in 0 1; do run mlp_b${m}_$rep "--workload mlp" GOCTR_MLP_SLAB_BLOCKED=$m; done
done
for m in 0 1; do run mlp100k_b$m "--workload mlp100k" GOCTR_MLP_SLAB_BLOCKED=$m; done
for m in 0 1; do
GOCTR_MLP_SLAB_BLOCKED=$m GOCTR_DBG=mlp timeout 120 python - <<'P' 2>&1 | grep "mlp_reduce" | tail -3
import sys, os
sys.path.insert(0, os.getcwd())
os.environ["GOCTR_NO_GRAPH"] = "1"
import numpy as np
from goctr_amd import capi, mlp as gmlp
capi.init(0)
rng = np.random.default_rng(1)
X = rng.random((1 << 16, 281), dtype=np.float32); y = (rng.random(1 << 16) < 0.5).astype(np.float32)
clf = gmlp.MLPClassifier([100], "relu", "adam", 1e-5); clf.BatchSize = 4096
clf.create([281, 100, 1], 4096, clf.init_params([281, 100, 1], rng)); clf.upload(X, y)
clf.train_steps(8); capi.sync()
P
done
