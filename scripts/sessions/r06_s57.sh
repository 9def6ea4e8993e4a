#!/bin/bash
# round 6, session 57: phase stamps of mlp_reduce_update_kernel (GOCTR_DBG=mlp), with and without the prefetch blocks
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for m in 1 0; do
echo "GOCTR_MLP_PREFETCH=$m"
GOCTR_MLP_PREFETCH=$m GOCTR_DBG=mlp timeout 120 python - <<'P' 2>&1 | grep "mlp_reduce\|mlp_chain wave 0" | tail -6
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
