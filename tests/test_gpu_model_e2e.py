"""The reference's ONLY end-to-end test of the DIN / YouTube-DNN path, on the device, with the oracle run beside it.

model/model_test.go:18-160 (TestMultiModel): dims (5, 3, 7, 7, 5), 100 000 learnable rows, batch 200, 20 epochs = 10 000 Adam
steps; model.Train (DIN: no early stop; YouTube-DNN: early stop 10) -> Marshal -> New*FromJson -> InitForwardOnlyVm(20) ->
Predict(118 rows: 5 whole batches of 20 and a padded one of 18) -> RocAuc32 > 0.5.  The reference asserts the AUC only; here the
same run goes through the host mirror (goctr_amd.model: every number comes out of libgoctr_hip.so) AND through the oracle from the
same initial weights, rows and dropout stream, and the two trajectories are compared -- the 10 000-step drift measurement the
shorter parity tests (12 / 20 steps) do not give.

The generator is the reference's rule (model_test.go:44-76) on numpy's stream (Go's math/rand stream cannot be reproduced without
Go): user profile, context, item features and the SECOND behaviour slot are U[0,1), the other two slots stay zero;
label = round(0.6 * (mean |profile - context| + mean |behaviour[1] - item|)).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIMS_E2E = (5, 3, 7, 5)          # uProfileDim, uBehaviorSize, uBehaviorDim (= iFeatureDim), cFeatureDim  (model_test.go:22-27)


def reference_test_data(n, seed):
    U, T, D, Cc = DIMS_E2E
    rng = np.random.default_rng(seed)
    X = np.zeros((n, U + T * D + D + Cc), np.float32)
    X[:, :U] = rng.random((n, U), dtype=np.float32)                               # :45-47
    X[:, U + T * D + D:] = rng.random((n, Cc), dtype=np.float32)                  # :48-50
    X[:, U + D:U + 2 * D] = rng.random((n, D), dtype=np.float32)                  # :51-53  (behaviour slot 1 of 0..2)
    X[:, U + T * D:U + T * D + D] = rng.random((n, D), dtype=np.float32)          # :54-56
    d1 = np.zeros(n, np.float32)
    for j in range(U):                                                            # :66-69 (float32 accumulation, in j order)
        d1 += np.abs(X[:, j] - X[:, U + T * D + D + j])
    d2 = np.zeros(n, np.float32)
    for j in range(D):                                                            # :72-74
        d2 += np.abs(X[:, U + D + j] - X[:, U + T * D + j])
    lab = (d1 / np.float32(U) + d2 / np.float32(D)) * np.float32(0.6)
    Y = np.floor(lab.astype(np.float64) + 0.5).astype(np.float32)                 # :75 math.Round (half away from zero; lab >= 0)
    return X, Y


# ---- tolerances, and where they come from (profiles/r06_e2e_drift.txt: scripts/e2e_model_test_probe.py on an MI355X, -------------
# scripts/e2e_oracle_sensitivity.py on the CPU) ------------------------------------------------------------------------------
# Device and oracle run the same arithmetic in different float32 summation orders (MFMA k-order, slab sums, the 6-product bf16
# split against the oracle's sequential loops): ~1e-7 relative per step.  Over 10 000 Adam steps from the reference's N(0,1) init
# (saturated sigmoids, gradients that pass through zero) such differences do not stay 1e-7: Adam turns a gradient entry within
# rounding noise of zero into a +-lr update.  MEASURED: the ORACLE against itself after a ONE-ulp change of its initial weights
# moves its epoch costs by 3e-7 ... 2.3e-3 (six runs) and its weights by up to 0.08; the device against the oracle: 7e-7 ... 1.6e-3
# in five of six runs, 1.2e-2 in one (a trajectory whose weights end 4.1 apart).  Even the oracle binary gives different
# trajectories on different hosts (glibc picks expf / logf variants by CPU).  What is compared after 10 000 steps is therefore two
# members of one family of float32 runs.  The bar:
#   * tier 1 (what seed 1 measured on the MI355X boxes: 7e-6 DIN, 1.1e-4 YouTube-DNN; AUCs equal to four digits):
#       every epoch cost within COST_TOL_E2E = 2e-3 (the size of the oracle's own response to one ulp), AUC within 0.005;
#   * tier 2, only if tier 1 fails (a host whose libm puts seed 1 on a sensitive trajectory): the oracle is run twice more with
#     its initial W0 / W1 moved by one ulp; the device must then stay within 5 x the spread of that family (costs and AUC),
#     and the first two epochs (1 000 steps, before trajectories separate: <= 8.5e-5 in all six measured runs) within 2e-4.
COST_TOL_E2E = 2e-3
AUC_TOL_E2E = 0.005
EARLY_COST_TOL = 2e-4
FAMILY_FACTOR = 5.0


def _oracle_run(oracle, kind, seed, early_stop, X, Y, p, pert=0):
    U, T, D, Cc = DIMS_E2E
    om = oracle.CtrModel(oracle.DIN if kind == 0 else oracle.YOUTUBE, U, T, D, Cc).init_gaussian(np.random.default_rng(seed))
    if pert == 1:
        om.W0[:] = om.W0 * np.float32(1 + 2.0 ** -23)
    if pert == 2:
        om.W1[:] = om.W1 * np.float32(1 - 2.0 ** -23)
    w0 = (om.W0.copy(), om.W1.copy(), om.W2.copy(), om.att0.copy())
    costs = om.train(X, Y, batch=200, epochs=20, early_stop=early_stop, drop_mode=2, p0=p, p1=p, seed=42)
    return om, w0, costs


@pytest.mark.parametrize("kind,early_stop", [(0, 0), (1, 10)], ids=["din", "youtube_early_stop_10"])
def test_reference_model_test_end_to_end(oracle, kind, early_stop):
    from goctr_amd import model as gm
    from goctr_amd.recommend import SampleInfo
    U, T, D, Cc = DIMS_E2E
    n, B, epochs, n_test, test_B = 100_000, 200, 20, 118, 20                      # model_test.go:21-33
    X, Y = reference_test_data(n, 42)
    si = SampleInfo.from_dims(U, T, D, Cc)
    dm = (gm.DinNet if kind == 0 else gm.YoutubeDnn)(U, T, D, D, Cc)              # NewDinNet / NewYoutubeDnn (:80, :119)
    oracle.set_threads(min(16, len(os.sched_getaffinity(0))))                     # (row-parallel loops: no summation order depends on it)
    try:
        om, (W0, W1, W2, att0), ref = _oracle_run(oracle, kind, 1, early_stop, X, Y, dm.d0)
        dm.set_weights("mlp0", W0); dm.set_weights("mlp1", W1); dm.set_weights("mlp2", W2)
        if kind == 0:
            dm.set_weights("att0", att0)
        costs = gm.Train(U, T, D, D, Cc, n, B, epochs, early_stop, si, X, Y.reshape(-1, 1), dm, dropout_seed=42)     # :82-88 / :121-127
        assert len(costs) == len(ref), "device and oracle stopped at different epochs"
        assert np.all(np.isfinite(costs))
        blob = dm.Marshal()                                                           # :93 / :132
        dp = (gm.NewDinNetFromJson if kind == 0 else gm.NewYoutubeDnnFromJson)(blob)  # :96 / :135
        gm.InitForwardOnlyVm(U, T, D, D, Cc, test_B, dp)                              # :101 / :140
        y = gm.Predict(dp, n_test, test_B, si, X)                                     # :103 / :142
        assert y.shape == (n_test,) and np.all(np.isfinite(y))
        auc = oracle.roc_auc32(y, Y[:n_test])                                         # :108 / :147 utils.RocAuc32
        assert auc > 0.5                                                              # the reference's own assertion
        auc_ref = oracle.roc_auc32(om.predict(X[:n_test], test_B), Y[:n_test])
        diff = np.abs(costs - ref)
        if diff.max() <= COST_TOL_E2E and abs(auc - auc_ref) <= AUC_TOL_E2E:
            return                                                                    # tier 1
        # tier 2: the oracle's own one-ulp family on THIS host as the yard-stick
        fam_c, fam_a = [ref], [auc_ref]
        for pert in (1, 2):
            op, _, cp = _oracle_run(oracle, kind, 1, early_stop, X, Y, dm.d0, pert)
            assert len(cp) == len(ref)
            fam_c.append(cp); fam_a.append(oracle.roc_auc32(op.predict(X[:n_test], test_B), Y[:n_test]))
        fam_c = np.stack(fam_c)
        spread = fam_c.max(0) - fam_c.min(0)
        assert np.all(diff[:2] <= EARLY_COST_TOL), diff[:2]
        assert np.all(diff <= FAMILY_FACTOR * spread.max() + 1e-5), (diff, spread)
        assert abs(auc - auc_ref) <= max(AUC_TOL_E2E, FAMILY_FACTOR * (max(fam_a) - min(fam_a))), (auc, fam_a)
    finally:
        oracle.set_threads(1)
