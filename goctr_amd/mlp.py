"""Host mirror of the sklearn-port MLP used by go-ctr's "simple 2-layer MLP" path.

Reference: nn/neural_network/multilayer_perceptron.go:74-125 (MLPClassifier, NewMLPClassifier, Fit,
Predict), nn/neural_network/basemlp64.go (hyper-parameters :25-51, defaults :228-254, init :408-482,
validateHyperparameters :625-673) and the adapters model/mlp/mlp.go:15-65 (SimpleMlpFitWrap /
SimpleMlpPredWrap).  Arithmetic is float64 on the device (include/goctr.h, goctr_mlp_*); this file keeps
the reference's names, defaults and error behaviour and owns what is host-side in the reference too:
the init RNG (Q8: one-sided U[0,bound)) and the per-epoch shuffle.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import capi
from .recommend import Fitter, PredictAbstract, TrainSample

_ACT = {"identity": 0, "logistic": 1, "tanh": 2, "relu": 3}
_SOLVER = {"sgd": 0, "adam": 1}


class MLPClassifier:
    """nn.MLPClassifier (multilayer_perceptron.go:74-90) with BaseMultilayerPerceptron64's exported fields."""

    def __init__(self, hiddenLayerSizes, activation="relu", solver="adam", Alpha=1e-4):
        # NewBaseMultilayerPerceptron64 defaults (basemlp64.go:228-254)
        self.HiddenLayerSizes = list(hiddenLayerSizes)
        self.Activation, self.Solver, self.Alpha = activation, solver, Alpha
        self.BatchSize = 200
        self.LearningRate = "constant"
        self.LearningRateInit = 0.001
        self.MaxIter = 200
        self.Shuffle = True
        self.RandomState = None          # numpy Generator; None => time-seeded like basemlp64.go:446-448
        self.Tol = 1e-4
        self.Verbose = False
        self.Momentum, self.NesterovsMomentum = 0.9, True
        self.Beta1, self.Beta2, self.Epsilon = 0.9, 0.999, 1e-8
        self.NIterNoChange = 10
        self.BatchNormalize = False
        self.WeightDecay = 0.0
        # fitted state
        self.LossCurve = []
        self.NIter = 0
        self._h = None
        self._units = None

    # validateHyperparameters (basemlp64.go:625-673) panics; here: ValueError
    def _validate(self):
        if self.Activation not in _ACT:
            raise ValueError(f"The activation {self.Activation} is not supported.")
        if self.Solver not in _SOLVER:
            raise ValueError(f"The solver {self.Solver} is not supported (lbfgs has no device path).")
        if self.Alpha < 0 or self.LearningRateInit <= 0 or self.MaxIter <= 0:
            raise ValueError("invalid hyper-parameters")
        if any(s <= 0 for s in self.HiddenLayerSizes):
            raise ValueError(f"hiddenLayerSizes must be > 0, got {self.HiddenLayerSizes}.")
        if self.LearningRate != "constant":
            raise ValueError("only the constant learning-rate schedule has a device path")

    def _cfg(self, units, batch):
        c = capi.MlpCfg()
        capi.load().goctr_mlp_cfg_default(C.byref(c))
        c.n_layers = len(units)
        for i, u in enumerate(units):
            c.units[i] = u
        c.activation, c.solver, c.alpha = _ACT[self.Activation], _SOLVER[self.Solver], self.Alpha
        c.lr_init, c.beta1, c.beta2, c.eps = self.LearningRateInit, self.Beta1, self.Beta2, self.Epsilon
        c.momentum, c.nesterov = self.Momentum, int(self.NesterovsMomentum)
        c.batch_normalize, c.weight_decay = int(self.BatchNormalize), self.WeightDecay
        c.batch, c.max_iter, c.n_iter_no_change, c.tol = batch, self.MaxIter, self.NIterNoChange, self.Tol
        return c

    def init_params(self, units, rng):
        """initialize (basemlp64.go:432-475): packed [b_i | W_i], each = U[0,1) * sqrt(f/(fanIn+fanOut)),
        f = 2 for logistic else 6 -- one-sided on purpose (quirk Q8)."""
        theta = []
        for i in range(len(units) - 1):
            fi, fo = units[i], units[i + 1]
            bound = math.sqrt((2.0 if self.Activation == "logistic" else 6.0) / (fi + fo))
            theta.append(rng.random(fo + fi * fo) * bound)
        return np.concatenate(theta)

    def create(self, units, batch, theta):
        capi.init()
        self.close()
        self._units = list(units)
        self._h = C.c_void_p()
        cfg = self._cfg(units, batch)
        capi.check(capi.load().goctr_mlp_create(C.byref(cfg), C.byref(self._h)))
        self.set_params(theta)

    def set_params(self, theta):
        theta = np.ascontiguousarray(theta, np.float64)
        capi.check(capi.load().goctr_mlp_set_params(self._h, capi.ptr(theta, C.c_double), C.c_size_t(theta.size)))

    def get_params(self):
        n = capi.load().goctr_mlp_nparams(self._h)
        theta = np.empty(n, np.float64)
        capi.check(capi.load().goctr_mlp_get_params(self._h, capi.ptr(theta, C.c_double), C.c_size_t(n)))
        return theta

    def loss_grad(self, X, Y):
        X = np.ascontiguousarray(X, np.float64)
        Y = np.ascontiguousarray(Y, np.float64).reshape(X.shape[0], -1)
        g = np.empty(capi.load().goctr_mlp_nparams(self._h), np.float64)
        loss = C.c_double(0)
        capi.check(capi.load().goctr_mlp_loss_grad(self._h, capi.ptr(X, C.c_double), capi.ptr(Y, C.c_double),
                                                   C.c_int(X.shape[0]), C.byref(loss), capi.ptr(g, C.c_double)))
        return loss.value, g

    def Fit(self, X, Y, theta0=None, perm=None):
        """Base64.Fit (basemlp64.go:578-599) -> fit :484 -> fitStochastic :729.  X float32 rows (the adapter
        widens them, mlp.go:46-53), Y in {0,1}.  Every row is trained: a sample count that is not a multiple of the batch
        ends each epoch with one short batch, computed the reference's way (quirk Q11, basemlp64.go:790-812 -- its hidden
        block and output deltas keep the previous batch's rows beyond the short batch; csrc/mlp.hip goctr_mlp_fit)."""
        self._validate()
        X = capi.f32(X)
        Y = capi.f32(Y).reshape(X.shape[0], -1)
        rng = self.RandomState or np.random.default_rng()
        units = [X.shape[1], *self.HiddenLayerSizes, Y.shape[1]]
        batch = min(self.BatchSize if self.BatchSize > 0 else 200, X.shape[0])     # basemlp64.go:516-527
        rows = X.shape[0]
        theta = theta0 if theta0 is not None else self.init_params(units, rng)
        self.create(units, batch, theta)
        if perm is None and self.Shuffle:
            # the reference shuffles X,Y in place every epoch (cumulative); equivalent index form
            order = np.arange(rows)
            perms = []
            for _ in range(self.MaxIter):
                order = order[rng.permutation(rows)]
                perms.append(order.copy())
            perm = np.stack(perms).astype(np.int32)
        curve = np.zeros(self.MaxIter, np.float64)
        ran = C.c_int(0)
        pp = np.ascontiguousarray(perm, np.int32) if perm is not None else None
        capi.check(capi.load().goctr_mlp_fit(self._h, capi.ptr(X, C.c_float), capi.ptr(Y, C.c_float), C.c_int64(rows),
                                             capi.ptr(pp, C.c_int32), capi.ptr(curve, C.c_double), C.byref(ran)))
        return self._fitted(curve, ran.value)

    def _fitted(self, curve, ran):
        self.NIter = ran
        self.LossCurve = curve[:ran].tolist()
        if self.Verbose:
            for i, l in enumerate(self.LossCurve):
                print("Iteration %d, loss = %.8f" % (i + 1, l))
        return self

    def FitResident(self, perm=None):
        """fitStochastic over the rows a previous upload() left in HBM (goctr_mlp_fit_resident: what Fit runs after its
        upload).  perm: [MaxIter][rows] int32 row order per epoch, or None = the given order every epoch."""
        curve = np.zeros(self.MaxIter, np.float64)
        ran = C.c_int(0)
        pp = np.ascontiguousarray(perm, np.int32) if perm is not None else None
        capi.check(capi.load().goctr_mlp_fit_resident(self._h, capi.ptr(pp, C.c_int32), capi.ptr(curve, C.c_double), C.byref(ran)))
        return self._fitted(curve, ran.value)

    def Predict(self, X):
        """predictProbas (basemlp64.go:897) for a binary classifier: probabilities, float32 like mlp.go:33-38"""
        X = capi.f32(X)
        no = self._units[-1]
        y = np.empty((X.shape[0], no), np.float32)
        capi.check(capi.load().goctr_mlp_predict(self._h, capi.ptr(X, C.c_float), C.c_int64(X.shape[0]),
                                                 capi.ptr(y, C.c_float)))
        return y

    def upload(self, X, Y):
        X = capi.f32(X)
        Y = capi.f32(Y).reshape(X.shape[0], -1)
        capi.check(capi.load().goctr_mlp_upload(self._h, capi.ptr(X, C.c_float), capi.ptr(Y, C.c_float),
                                                C.c_int64(X.shape[0])))

    def train_steps(self, n_steps, first_batch=0):
        capi.check(capi.load().goctr_mlp_train_steps(self._h, C.c_int64(first_batch), C.c_int(n_steps)))

    def close(self):
        if self._h:
            capi.load().goctr_mlp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def NewMLPClassifier(hiddenLayerSizes, activation, solver, Alpha):
    """multilayer_perceptron.go:81-90"""
    return MLPClassifier(hiddenLayerSizes, activation, solver, Alpha)


class SimpleMlpPredWrap(PredictAbstract):
    """model/mlp/mlp.go:11-39"""

    def __init__(self, pred: MLPClassifier):
        self.pred = pred

    def Predict(self, X):
        return self.pred.Predict(X)


class SimpleMlpFitWrap(Fitter):
    """model/mlp/mlp.go:41-65"""

    def __init__(self, Model: MLPClassifier):
        self.Model = Model

    def Fit(self, trainSample: TrainSample) -> PredictAbstract:
        X = np.asarray(trainSample.X, np.float32).reshape(trainSample.Rows, trainSample.XCols)
        self.Model.Fit(X, np.asarray(trainSample.Y, np.float32))
        return SimpleMlpPredWrap(self.Model)
