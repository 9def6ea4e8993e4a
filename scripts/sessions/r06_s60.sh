#!/bin/bash
# round 6, session 60: mlp_reduce_update_kernel with the block's (not the lane's) layer descriptor, the state load consumed late, the loss
# block's requests in one round: tests, A/B against HEAD's library (libgoctr_hip_old.so) on one box, stamps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R; O=gpurun_out/r06_s60; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_fullsize.py tests/test_gpu_multi.py tests/test_gpu_comm.py tests/test_gpu_resume.py -q -m gpu -x -p no:cacheprovider -k "mlp or Mlp or flagship or sklearn" 2>&1 | tail -2
run() {  # name, workload args, env...
  n=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --steps 200 --warmup 20 --no-cpu-baseline > $O/$n.json 2> $O/$n.err
  python - <<P
import json
d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['value'], d['ms_per_step'], d.get('us_per_update'), d.get('timed_regions_ms')[1:6])
P
}
for rep in 1 2 3; do
run mlp_old_$rep "--workload mlp" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
run mlp_new_$rep "--workload mlp"
done
run mlp100k_old "--workload mlp100k" GOCTR_LIB=$R/goctr_amd/libgoctr_hip_old.so
run mlp100k_new "--workload mlp100k"
GOCTR_DBG=mlp timeout 120 python - <<'P' 2>&1 | grep "mlp_reduce" | tail -3
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
