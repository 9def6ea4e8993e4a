"""How far does the ORACLE move when its initial weights move by one float32 ulp?  (CPU only.)  The yard-stick for
tests/test_gpu_model_e2e.py: two float32 evaluations of the same 10 000-step run (model/model_test.go:18-160) that differ in
rounding only -- here: W0 scaled by (1 + 2^-23), i.e. every entry moved by <= 1 ulp -- separate by this much; the device's
summation order differs from the oracle's by the same kind of amount per step."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_model_e2e import reference_test_data, DIMS_E2E   # noqa: E402
from oracle import pyoracle as o                                       # noqa: E402

U, T, D, Cc = DIMS_E2E
X, Y = reference_test_data(100000, 42)
o.set_threads(min(16, len(os.sched_getaffinity(0))))
for seed in (1, 2, 3):
    for kind, es, p in ((0, 0, 0.005), (1, 10, 0.003)):
        runs = []
        for pert in (0, 1, 2):
            m = o.CtrModel(o.DIN if kind == 0 else o.YOUTUBE, U, T, D, Cc).init_gaussian(np.random.default_rng(seed))
            if pert == 1:
                m.W0[:] = m.W0 * np.float32(1 + 2.0 ** -23)
            if pert == 2:
                m.W1[:] = m.W1 * np.float32(1 - 2.0 ** -23)
            c = m.train(X, Y, batch=200, epochs=20, early_stop=es, drop_mode=2, p0=p, p1=p, seed=42)
            y = m.predict(X[:118], 20)
            runs.append((c, m.W0.copy(), o.roc_auc32(y, Y[:118])))
        for k in (1, 2):
            n = min(len(runs[0][0]), len(runs[k][0]))
            d = np.abs(runs[0][0][:n] - runs[k][0][:n])
            print(f"seed {seed} kind {kind} perturbation {k}: epochs {len(runs[k][0])}/{len(runs[0][0])} cost diff max {d.max():.2e} (last epoch {d[-1]:.2e}) | "
                  f"W0 diff {np.max(np.abs(runs[0][1] - runs[k][1])):.3g} | AUC {runs[k][2]:.4f} vs {runs[0][2]:.4f}", flush=True)
