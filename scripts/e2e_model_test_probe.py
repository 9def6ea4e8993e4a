"""How far do the device and the oracle drift apart over the reference's own end-to-end test (model/model_test.go:18-160:
100 000 rows x 20 epochs at batch 200 = 10 000 Adam steps)?  Prints what tests/test_gpu_model_e2e.py's tolerances are derived
from: per-epoch cost differences (the cost of an epoch is its LAST batch's, model.go:186-199), weight differences, AUCs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_model_e2e import reference_test_data, DIMS_E2E   # noqa: E402
from goctr_amd import capi, model as gm                                # noqa: E402
from goctr_amd.recommend import SampleInfo                             # noqa: E402
from oracle import pyoracle as o                                       # noqa: E402

capi.init(0)
U, T, D, Cc = DIMS_E2E
X, Y = reference_test_data(int(os.environ.get("E2E_ROWS", "100000")), 42)
si = SampleInfo.from_dims(U, T, D, Cc)
o.set_threads(min(16, len(os.sched_getaffinity(0))))
for seed in (1, 2, 3):
    for kind, es in ((0, 0), (1, 10)):
        om = o.CtrModel(o.DIN if kind == 0 else o.YOUTUBE, U, T, D, Cc).init_gaussian(np.random.default_rng(seed))
        dm = (gm.DinNet if kind == 0 else gm.YoutubeDnn)(U, T, D, D, Cc)
        dm.set_weights("mlp0", om.W0); dm.set_weights("mlp1", om.W1); dm.set_weights("mlp2", om.W2)
        if kind == 0:
            dm.set_weights("att0", om.att0)
        t0 = time.time()
        costs = gm.Train(U, T, D, D, Cc, X.shape[0], 200, 20, es, si, X, Y.reshape(-1, 1), dm, dropout_seed=42)
        t1 = time.time()
        ref = om.train(X, Y, batch=200, epochs=20, early_stop=es, drop_mode=2, p0=dm.d0, p1=dm.d1, seed=42)
        t2 = time.time()
        n = min(len(costs), len(ref))
        diff = np.abs(costs[:n] - ref[:n])
        dp = (gm.NewDinNetFromJson if kind == 0 else gm.NewYoutubeDnnFromJson)(dm.Marshal())
        gm.InitForwardOnlyVm(U, T, D, D, Cc, 20, dp)
        yd = gm.Predict(dp, 118, 20, si, X)
        yo = om.predict(X[:118], 20)
        wd = {nm: float(np.max(np.abs(dm.get_weights(nm) - w))) for nm, w in (("mlp0", om.W0), ("mlp1", om.W1), ("mlp2", om.W2))}
        print(f"seed {seed} kind {kind}: epochs {len(costs)}/{len(ref)} device {t1 - t0:.2f}s oracle {t2 - t1:.1f}s | cost diff per epoch "
              f"{np.array2string(diff, precision=2)} max {diff.max():.2e} | weights {wd} | AUC {o.roc_auc32(yd, Y[:118]):.4f} vs "
              f"{o.roc_auc32(yo, Y[:118]):.4f} | max |y - y_oracle| {np.max(np.abs(yd - yo)):.2e}", flush=True)
