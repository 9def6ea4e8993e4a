"""ctypes binding of the CPU oracle (oracle/libgoctr_oracle.so).

TEST INFRASTRUCTURE ONLY (see oracle/goctr_oracle.h): imported by tests/, by
``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline`` leg -- never by the product
package ``goctr_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgoctr_oracle.so")


def build(force: bool = False) -> str:
    srcs = ["orc_ops.c", "orc_ctr.c", "orc_sklmlp.c", "orc_w2v.c", "goctr_oracle.h", "Makefile"]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, s)) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _proto(_lib)
    return _lib


def _p(a, ty):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ty))


f32p, f64p, i32p, i64p, u8p = (C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                               C.POINTER(C.c_int64), C.POINTER(C.c_uint8))


class CtrCfg(C.Structure):
    _fields_ = [("kind", C.c_int), ("att", C.c_int), ("U", C.c_int), ("T", C.c_int), ("D", C.c_int),
                ("C", C.c_int), ("H1", C.c_int), ("H2", C.c_int)]


class CtrWeights(C.Structure):
    _fields_ = [("W0", f32p), ("W1", f32p), ("W2", f32p), ("att0", f32p)]


class AdamState(C.Structure):
    _fields_ = [(n, f32p) for n in ("m0", "v0", "m1", "v1", "m2", "v2", "ma", "va")] + [("iter", C.c_int)]


class AdamCfg(C.Structure):
    _fields_ = [("lr", C.c_double), ("l2", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_double), ("adam_div_by_batch", C.c_int), ("adam_l2_before_batch_div", C.c_int)]


class Dropout(C.Structure):
    _fields_ = [("mode", C.c_int), ("p0", C.c_float), ("p1", C.c_float), ("m0", f32p), ("m1", f32p),
                ("seed", C.c_uint32), ("step", C.c_uint32)]


class MlpCfg(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("units", C.c_int * 8), ("activation", C.c_int), ("alpha", C.c_double),
                ("batch_normalize", C.c_int), ("weight_decay", C.c_double)]


class MlpOpt(C.Structure):
    _fields_ = [("solver", C.c_int), ("lr_init", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_double), ("momentum", C.c_double), ("nesterov", C.c_int), ("t", C.c_double),
                ("beta1t", C.c_double), ("beta2t", C.c_double), ("ms", f64p), ("vs", f64p),
                ("velocities", f64p), ("lr", C.c_double)]


class Lcg(C.Structure):
    _fields_ = [("next", C.c_uint64)]


class W2vCfg(C.Structure):
    _fields_ = [("dim", C.c_int), ("window", C.c_int), ("optimizer", C.c_int), ("model", C.c_int),
                ("neg_samples", C.c_int), ("init_lr", C.c_double), ("min_lr", C.c_double),
                ("update_lr_batch", C.c_int64), ("max_depth", C.c_int)]


def _proto(L):
    L.orc_bce32.restype = C.c_float
    L.orc_mse32.restype = C.c_float
    L.orc_rms32.restype = C.c_float
    L.orc_roc_auc.restype = C.c_double
    L.orc_roc_auc32.restype = C.c_float
    L.orc_dropout_keep.restype = C.c_float
    L.orc_dropout_keep.argtypes = [C.c_uint32] * 5 + [C.c_float]
    L.orc_ctr_loss_grad.restype = C.c_float
    L.orc_mlp_nparams.restype = C.c_size_t
    L.orc_mlp_loss_grad.restype = C.c_double
    L.orc_sigmoid_lookup.restype = C.c_double
    L.orc_sigmoid_lookup.argtypes = [f64p, C.c_double]
    L.orc_subsample_keep.restype = C.c_double
    L.orc_subsample_keep.argtypes = [C.c_double, C.c_int64]
    L.orc_huffman_paths.restype = C.c_int64
    L.orc_huffman_paths_slow.restype = C.c_int64
    L.orc_prelu32.argtypes = [f32p, C.c_float, f32p, C.c_int]


def set_threads(n: int):
    lib().orc_set_threads(C.c_int(n))


# ----------------------------------------------------------------- ops --
def prelu32(x, slope):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().orc_prelu32(_p(x, C.c_float), C.c_float(slope), _p(out, C.c_float), C.c_int(x.size))
    return out


def _bcast3(x, y):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    if x.ndim != y.ndim:
        return None
    if x.ndim == 2:
        if x.shape != y.shape:
            return None
        return x.reshape(x.shape[0], 1, -1), y.reshape(y.shape[0], 1, -1), (x.shape[0],)
    B, Tx, D = x.shape
    B2, Ty, D2 = y.shape
    if B != B2 or D != D2:
        return None
    return x, y, (B, max(Tx, Ty))


def cosine_similarity(x, y):
    r = _bcast3(x, y)
    if r is None:
        raise ValueError("x, y shapes not supported")
    x3, y3, oshape = r
    out = np.empty(x3.shape[0] * max(x3.shape[1], y3.shape[1]), np.float32)
    rc = lib().orc_cosine_similarity(_p(x3, C.c_float), C.c_int(x3.shape[1]), _p(y3, C.c_float),
                                     C.c_int(y3.shape[1]), C.c_int(x3.shape[0]), C.c_int(x3.shape[2]),
                                     _p(out, C.c_float))
    if rc != 0:
        raise ValueError("x, y shapes not supported")
    return out.reshape(oshape)


def euc_distance(x, y):
    r = _bcast3(x, y)
    if r is None:
        raise ValueError("x, y shapes not supported")
    x3, y3, oshape = r
    out = np.empty(x3.shape[0] * max(x3.shape[1], y3.shape[1]), np.float32)
    rc = lib().orc_euc_distance(_p(x3, C.c_float), C.c_int(x3.shape[1]), _p(y3, C.c_float),
                                C.c_int(y3.shape[1]), C.c_int(x3.shape[0]), C.c_int(x3.shape[2]),
                                _p(out, C.c_float))
    if rc != 0:
        raise ValueError("x, y shapes not supported")
    return out.reshape(oshape)


def _cost(fn, p, y):
    p = np.ascontiguousarray(p, np.float32).ravel()
    y = np.ascontiguousarray(y, np.float32).ravel()
    return float(fn(_p(p, C.c_float), _p(y, C.c_float), C.c_int(p.size)))


def bce32(p, y):
    return _cost(lib().orc_bce32, p, y)


def mse32(p, y):
    return _cost(lib().orc_mse32, p, y)


def rms32(p, y):
    return _cost(lib().orc_rms32, p, y)


def roc_auc(score, y):
    s = np.ascontiguousarray(score, np.float64).ravel()
    t = np.ascontiguousarray(y, np.float64).ravel()
    return float(lib().orc_roc_auc(_p(s, C.c_double), _p(t, C.c_double), C.c_int(s.size)))


def roc_auc32(score, y):
    s = np.ascontiguousarray(score, np.float32).ravel()
    t = np.ascontiguousarray(y, np.float32).ravel()
    return float(lib().orc_roc_auc32(_p(s, C.c_float), _p(t, C.c_float), C.c_int(s.size)))


def roc_curve(score, y, pos_label):
    s = np.ascontiguousarray(score, np.float64).ravel()
    t = np.ascontiguousarray(y, np.float64).ravel()
    n = s.size
    fpr, tpr, thr = (np.zeros(n + 2) for _ in range(3))
    m = lib().orc_roc_curve(_p(s, C.c_double), _p(t, C.c_double), C.c_double(pos_label), C.c_int(n),
                            _p(fpr, C.c_double), _p(tpr, C.c_double), _p(thr, C.c_double))
    return fpr[:m], tpr[:m], thr[:m]


# ------------------------------------------------------------ assembly --
def assemble_rows(emb, ub_ids, item_ids, user_feat, item_feat):
    ub_ids = np.ascontiguousarray(ub_ids, np.int32)
    item_ids = np.ascontiguousarray(item_ids, np.int32)
    user_feat = np.ascontiguousarray(user_feat, np.float32)
    item_feat = np.ascontiguousarray(item_feat, np.float32)
    rows, T = ub_ids.shape
    U, Cc = user_feat.shape[1], item_feat.shape[1]
    if emb is not None:
        emb = np.ascontiguousarray(emb, np.float32)
        V, D = emb.shape
    else:
        raise ValueError("emb required (pass zeros for the no-ItemEmbedding case)")
    X = np.empty((rows, U + T * D + D + Cc), np.float32)
    lib().orc_assemble_rows(_p(emb, C.c_float), C.c_int64(V), C.c_int(D), C.c_int(T), _p(ub_ids, C.c_int32),
                            _p(item_ids, C.c_int32), _p(user_feat, C.c_float), C.c_int(U),
                            _p(item_feat, C.c_float), C.c_int(Cc), C.c_int64(rows), _p(X, C.c_float))
    return X


# -------------------------------------------------------- DIN / YouTube --
DIN, YOUTUBE = 0, 1
ATT_COSINE, ATT_EUCLID = 0, 1


def ranges_for(U, T, D, Cc):
    """rcmd.go:401-422 SampleInfo column ranges as 8 ints."""
    a = U
    b = a + T * D
    c = b + D
    d = c + Cc
    return np.array([0, a, a, b, b, c, c, d], np.int32)


class CtrModel:
    """Host-side handle on oracle weights (numpy-owned)."""

    def __init__(self, kind, U, T, D, Cc, H1=200, H2=80, att=ATT_COSINE):
        self.cfg = CtrCfg(kind, att, U, T, D, Cc, H1, H2)
        self.I = U + 2 * D + Cc
        self.W0 = np.zeros((self.I, H1), np.float32)
        self.W1 = np.zeros((H1, H2), np.float32)
        self.W2 = np.zeros((H2, 1), np.float32)
        self.att0 = np.ones((T,), np.float32)
        self.ranges = ranges_for(U, T, D, Cc)
        self.xcols = U + T * D + D + Cc

    def init_gaussian(self, rng):
        """din.go:187-191: N(0,1) weights, att0 = 1 (Q6)."""
        self.W0[:] = rng.standard_normal(self.W0.shape).astype(np.float32)
        self.W1[:] = rng.standard_normal(self.W1.shape).astype(np.float32)
        self.W2[:] = rng.standard_normal(self.W2.shape).astype(np.float32)
        self.att0[:] = 1.0
        return self

    def _w(self):
        return CtrWeights(_p(self.W0, C.c_float), _p(self.W1, C.c_float), _p(self.W2, C.c_float),
                          _p(self.att0, C.c_float))

    def _drop(self, drop):
        if drop is None:
            return None, None
        d = Dropout()
        d.mode = drop.get("mode", 0)
        d.p0 = drop.get("p0", 0.0)
        d.p1 = drop.get("p1", 0.0)
        keep = []
        for k in ("m0", "m1"):
            m = drop.get(k)
            if m is not None:
                m = np.ascontiguousarray(m, np.float32)
                keep.append(m)
                setattr(d, k, _p(m, C.c_float))
        d.seed = drop.get("seed", 0)
        d.step = drop.get("step", 0)
        return d, keep

    def emb_loss_grad(self, E, ub_ids, item_ids, user_feat, ctx_feat, Y, B=None, drop=None, want_grad=True):
        """trainable-embedding EXTENSION (orc_embtrain.c): float64 (loss, dE [V,D])"""
        E = np.ascontiguousarray(E, np.float64)
        ub_ids = np.ascontiguousarray(ub_ids, np.int32)
        item_ids = np.ascontiguousarray(item_ids, np.int32)
        uf = np.ascontiguousarray(user_feat, np.float32)
        cf = np.ascontiguousarray(ctx_feat, np.float32)
        Y = np.ascontiguousarray(Y, np.float32)
        valid = ub_ids.shape[0]
        B = B or valid
        dE = np.empty_like(E) if want_grad else None
        d, _keep = self._drop(drop)
        w = self._w()
        lib().orc_embtrain_loss_grad.restype = C.c_double
        loss = lib().orc_embtrain_loss_grad(C.byref(self.cfg), C.byref(w), _p(E, C.c_double), C.c_int64(E.shape[0]),
                                            _p(ub_ids, C.c_int32), _p(item_ids, C.c_int32), _p(uf, C.c_float),
                                            _p(cf, C.c_float), _p(Y, C.c_float), C.c_int(B), C.c_int(valid),
                                            C.byref(d) if d is not None else None, _p(dE, C.c_double))
        return (loss, dE) if want_grad else loss

    def forward(self, X, B=None, drop=None, want_internals=False):
        X = np.ascontiguousarray(X, np.float32)
        valid = X.shape[0]
        B = B or valid
        y = np.empty(B, np.float32)
        d, _keep = self._drop(drop)
        cfg = self.cfg
        outs = {}
        if want_internals:
            outs = dict(h0=np.empty((B, self.I), np.float32), A0=np.empty((B, cfg.H1), np.float32),
                        A1=np.empty((B, cfg.H2), np.float32), gate=np.empty((B, cfg.T), np.float32),
                        wgt=np.empty((B, cfg.T), np.float32))
        w = self._w()
        lib().orc_ctr_forward(C.byref(cfg), C.byref(w), _p(X, C.c_float), C.c_int(X.shape[1]),
                              _p(self.ranges, C.c_int32), C.c_int(B), C.c_int(valid),
                              C.byref(d) if d is not None else None, _p(y, C.c_float),
                              *[_p(outs.get(k), C.c_float) for k in ("h0", "A0", "A1", "gate", "wgt")])
        return (y, outs) if want_internals else y

    def loss_grad(self, X, Y, B=None, drop=None):
        X = np.ascontiguousarray(X, np.float32)
        Y = np.ascontiguousarray(Y, np.float32).ravel()
        valid = X.shape[0]
        B = B or valid
        g = dict(W0=np.zeros_like(self.W0), W1=np.zeros_like(self.W1), W2=np.zeros_like(self.W2),
                 att0=np.zeros_like(self.att0))
        gw = CtrWeights(*[_p(g[k], C.c_float) for k in ("W0", "W1", "W2", "att0")])
        y = np.empty(B, np.float32)
        d, _keep = self._drop(drop)
        w = self._w()
        cost = lib().orc_ctr_loss_grad(C.byref(self.cfg), C.byref(w), _p(X, C.c_float), C.c_int(X.shape[1]),
                                       _p(self.ranges, C.c_int32), _p(Y, C.c_float), C.c_int(B), C.c_int(valid),
                                       C.byref(d) if d is not None else None, C.byref(gw), _p(y, C.c_float))
        return float(cost), g, y

    def loss_grad_f64(self, X, Y, B=None):
        """FLOAT64 evaluation (numpy) of the same graph as orc_ctr.c's fwd_row / orc_ctr_loss_grad (din.go:219-323,
        dnn.go:162-184, activation.go:23-83, cost.go:9-17; no dropout) on the float32 inputs and weights: the truth
        that bounds the float32 summation-order error of both the float32 oracle and the device.  The BCE constant is
        float32(1 + 1e-8) == 1 like in the reference (quirk Q2).  Returns (cost, grads dict, y)."""
        c = self.cfg
        U, T, D, Cc = c.U, c.T, c.D, c.C
        X = np.asarray(X, np.float64)
        valid = X.shape[0]
        B = B or valid
        if B > valid:
            X = np.vstack([X, np.zeros((B - valid, X.shape[1]))])
        yv = np.zeros(B)
        yv[:valid] = np.asarray(Y, np.float64).ravel()
        W0, W1, W2, att0 = (np.asarray(a, np.float64) for a in (self.W0, self.W1, self.W2, self.att0))

        def sigm(x):                                  # gorgonia's float32 sigmoid clamps, evaluated in float64
            out = 1.0 / (1.0 + np.exp(-np.clip(x, -88.0, 15.0)))
            return np.where(x < -88.0, 0.0, np.where(x > 15.0, 1.0, out))

        u, ub = X[:, :U], X[:, U:U + T * D].reshape(B, T, D)
        v, cx = X[:, U + T * D:U + T * D + D], X[:, U + T * D + D:U + T * D + D + Cc]
        if c.kind == DIN:
            if c.att == ATT_COSINE:
                sxx, sxy = (ub * ub).sum(-1), (ub * v[:, None, :]).sum(-1)
                yn = np.sqrt((v * v).sum(-1))
                wv = (sxy / (np.sqrt(sxx) * yn[:, None] + 1e-8) + 1.0) / 2.0
            else:
                wv = 1.0 - np.sqrt(((ub - v[:, None, :]) ** 2).sum(-1))
            g = sigm(wv * att0[None, :])
        else:
            wv = np.zeros((B, T))
            g = np.ones((B, T))
        p = (g[..., None] * ub).sum(1) / T
        h0 = np.concatenate([u, p, v, cx], axis=1)
        A0 = sigm(h0 @ W0)
        A1 = sigm(A0 @ W1)
        y = sigm(A1 @ W2).ravel()
        with np.errstate(divide="ignore", invalid="ignore"):
            cost = -np.mean(yv * np.log(y) + (1.0 - yv) * np.log(1.0 - y))
            dy = -((yv / y) - ((1.0 - yv) / (1.0 - y))) / B
        d2 = dy * (y * (1.0 - y))
        dz1 = (d2[:, None] * W2.T) * (A1 * (1.0 - A1))
        dz0 = (dz1 @ W1.T) * (A0 * (1.0 - A0))
        grads = dict(W0=h0.T @ dz0, W1=A0.T @ dz1, W2=(A1.T @ d2).reshape(-1, 1), att0=np.zeros(T))
        if c.kind == DIN:
            dpT = dz0 @ W0[U:U + D].T / T
            dg = (dpT[:, None, :] * ub).sum(-1)
            grads["att0"] = (dg * (g * (1.0 - g)) * wv).sum(0)
        return float(cost), grads, y

    def adam_step(self, grads, state=None, adam=None, batch=1):
        """ONE gorgonia AdamSolver.Step (orc_ctr_adam_step) on this model's weights with the given float32 gradients
        (dict W0/W1/W2/att0; consumed: the solver zeroes them).  state: dict of moments + 'iter' (created when None)."""
        ac = adam or default_adam()
        if state is None:
            state = dict(iter=0)
            for k, w in (("0", self.W0), ("1", self.W1), ("2", self.W2), ("a", self.att0)):
                state["m" + k] = np.zeros_like(w)
                state["v" + k] = np.zeros_like(w)
        g = {k: np.ascontiguousarray(grads[k], np.float32).reshape(getattr(self, k).shape).copy() for k in ("W0", "W1", "W2", "att0")}
        gw = CtrWeights(*[_p(g[k], C.c_float) for k in ("W0", "W1", "W2", "att0")])
        st = AdamState(*[_p(state[n], C.c_float) for n in ("m0", "v0", "m1", "v1", "m2", "v2", "ma", "va")], state["iter"])
        w = self._w()
        lib().orc_ctr_adam_step(C.byref(self.cfg), C.byref(w), C.byref(gw), C.byref(st), C.byref(ac), C.c_int(batch))
        state["iter"] = st.iter
        return state

    def train(self, X, Y, batch, epochs, early_stop=0, adam=None, drop_mode=0, p0=0.0, p1=0.0, seed=0):
        X = np.ascontiguousarray(X, np.float32)
        Y = np.ascontiguousarray(Y, np.float32).ravel()
        ac = adam or default_adam()
        costs = np.zeros(epochs, np.float32)
        w = self._w()
        ran = lib().orc_ctr_train(C.byref(self.cfg), C.byref(w), _p(X, C.c_float), _p(Y, C.c_float),
                                  C.c_int64(X.shape[0]), C.c_int(X.shape[1]), _p(self.ranges, C.c_int32),
                                  C.c_int(batch), C.c_int(epochs), C.c_int(early_stop), C.byref(ac),
                                  C.c_int(drop_mode), C.c_float(p0), C.c_float(p1), C.c_uint32(seed),
                                  _p(costs, C.c_float))
        return costs[:ran]

    def predict(self, X, batch):
        X = np.ascontiguousarray(X, np.float32)
        y = np.empty(X.shape[0], np.float32)
        w = self._w()
        lib().orc_ctr_predict(C.byref(self.cfg), C.byref(w), _p(X, C.c_float), C.c_int64(X.shape[0]),
                              C.c_int(X.shape[1]), _p(self.ranges, C.c_int32), C.c_int(batch), _p(y, C.c_float))
        return y


def default_adam():
    """model.go:88: NewAdamSolver(WithLearnRate(0.01), WithBatchSize(B), WithL2Reg(0.0001))."""
    return AdamCfg(0.01, 0.0001, 0.9, 0.999, 1e-8, 1, 1)


# ------------------------------------------------------- sklearn-port MLP --
ACT = {"identity": 0, "logistic": 1, "tanh": 2, "relu": 3}
SOLVER = {"sgd": 0, "adam": 1}


def mlp_cfg(units, activation="relu", alpha=1e-4, batch_normalize=False, weight_decay=0.0):
    cfg = MlpCfg()
    cfg.n_layers = len(units)
    for i, u in enumerate(units):
        cfg.units[i] = u
    cfg.activation = ACT[activation]
    cfg.alpha = alpha
    cfg.batch_normalize = int(batch_normalize)
    cfg.weight_decay = weight_decay
    return cfg


def mlp_nparams(cfg):
    return int(lib().orc_mlp_nparams(C.byref(cfg)))


def mlp_predict(cfg, theta, X):
    X = np.ascontiguousarray(X, np.float64)
    theta = np.ascontiguousarray(theta, np.float64)
    out = np.empty((X.shape[0], cfg.units[cfg.n_layers - 1]), np.float64)
    lib().orc_mlp_predict(C.byref(cfg), _p(theta, C.c_double), _p(X, C.c_double), C.c_int(X.shape[0]),
                          _p(out, C.c_double))
    return out


def mlp_loss_grad(cfg, theta, X, Y):
    X = np.ascontiguousarray(X, np.float64)
    Y = np.ascontiguousarray(Y, np.float64)
    assert theta.dtype == np.float64 and theta.flags.c_contiguous
    g = np.zeros_like(theta)
    loss = lib().orc_mlp_loss_grad(C.byref(cfg), _p(theta, C.c_double), _p(X, C.c_double), _p(Y, C.c_double),
                                   C.c_int(X.shape[0]), _p(g, C.c_double))
    return float(loss), g


def mlp_blocks(cfg, B):
    """the activation / delta blocks fit allocates once per call (basemlp64.go:529-545): B rows per layer, zero-filled"""
    L = cfg.n_layers
    acts = [None] + [np.zeros((B, cfg.units[i]), np.float64) for i in range(1, L)]
    deltas = [np.zeros((B, cfg.units[i]), np.float64) for i in range(1, L)]
    return acts, deltas


def mlp_loss_grad_rows(cfg, theta, X, Y, B, acts, deltas):
    """backprop on the caller's blocks for a batch of ns = len(X) <= B rows; ns < B is the reference's short last batch
    (quirk Q11): rows [ns, B) of acts[1] / deltas[-1] are whatever the previous call left there"""
    X = np.ascontiguousarray(X, np.float64)
    Y = np.ascontiguousarray(Y, np.float64)
    assert theta.dtype == np.float64 and theta.flags.c_contiguous
    L = cfg.n_layers
    pa = (C.POINTER(C.c_double) * 8)()
    pd = (C.POINTER(C.c_double) * 8)()
    for i in range(1, L):
        assert acts[i].shape == (B, cfg.units[i]) and acts[i].flags.c_contiguous and acts[i].dtype == np.float64
        assert deltas[i - 1].shape == (B, cfg.units[i]) and deltas[i - 1].flags.c_contiguous
        pa[i] = _p(acts[i], C.c_double)
        pd[i - 1] = _p(deltas[i - 1], C.c_double)
    g = np.zeros_like(theta)
    f = lib().orc_mlp_loss_grad_rows
    f.restype = C.c_double
    loss = f(C.byref(cfg), _p(theta, C.c_double), _p(X, C.c_double), _p(Y, C.c_double), C.c_int(X.shape[0]), C.c_int(B),
             pa, pd, _p(g, C.c_double))
    return float(loss), g


class MlpOptimizer:
    def __init__(self, solver, nparams, lr_init=0.001):
        self.o = MlpOpt()
        self.n = nparams
        lib().orc_mlp_opt_init(C.byref(self.o), C.c_int(SOLVER[solver]), C.c_size_t(nparams))
        self.o.lr_init = lr_init
        self.o.lr = lr_init

    def update(self, theta, grads):
        grads = np.ascontiguousarray(grads, np.float64)
        lib().orc_mlp_update(C.byref(self.o), _p(theta, C.c_double), _p(grads, C.c_double), C.c_size_t(self.n))

    def __del__(self):
        try:
            lib().orc_mlp_opt_free(C.byref(self.o))
        except Exception:
            pass


def mlp_fit(cfg, theta, opt, X, Y, batch, max_iter, tol=1e-4, n_iter_no_change=10, perm=None):
    X = np.ascontiguousarray(X, np.float64)
    Y = np.ascontiguousarray(Y, np.float64)
    curve = np.zeros(max_iter, np.float64)
    if perm is not None:
        perm = np.ascontiguousarray(perm, np.int32)
    it = lib().orc_mlp_fit(C.byref(cfg), _p(theta, C.c_double), C.byref(opt.o), _p(X, C.c_double),
                           _p(Y, C.c_double), C.c_int64(X.shape[0]), C.c_int(batch), C.c_int(max_iter),
                           C.c_double(tol), C.c_int(n_iter_no_change), _p(perm, C.c_int32) if perm is not None else None,
                           _p(curve, C.c_double))
    return curve[:it]


# ---------------------------------------------------------------- item2vec --
def lcg_stream(n, value, seed=1):
    g = Lcg(seed)
    return [lib().orc_lcg_next(C.byref(g), C.c_int(value)) for _ in range(n)]


def sigmoid_table():
    t = np.empty(1000, np.float64)
    lib().orc_sigmoid_table(_p(t, C.c_double))
    return t


def sigmoid_lookup(table, x):
    return float(lib().orc_sigmoid_lookup(_p(table, C.c_double), C.c_double(x)))


def subsample_keep(threshold, count):
    return float(lib().orc_subsample_keep(C.c_double(threshold), C.c_int64(count)))


def index_per_thread(threads, n):
    out = np.zeros(threads + 1, np.int64)
    lib().orc_index_per_thread(C.c_int(threads), C.c_int64(n), _p(out, C.c_int64))
    return out


def huffman_paths(counts, max_depth=100, slow=False):
    counts = np.ascontiguousarray(counts, np.int64)
    V = counts.size
    off = np.zeros(V + 1, np.int64)
    fn = lib().orc_huffman_paths_slow if slow else lib().orc_huffman_paths
    total = fn(_p(counts, C.c_int64), C.c_int64(V), C.c_int(max_depth), _p(off, C.c_int64), None, None, C.c_int64(0))
    nodes = np.zeros(max(total, 1), np.int32)
    codes = np.zeros(max(total, 1), np.uint8)
    fn(_p(counts, C.c_int64), C.c_int64(V), C.c_int(max_depth), _p(off, C.c_int64), _p(nodes, C.c_int32),
       _p(codes, C.c_uint8), C.c_int64(total))
    return off, nodes[:total], codes[:total]


def w2v_cfg(dim=16, window=5, optimizer="hs", model="skipgram", neg_samples=5, init_lr=0.025,
            min_lr=0.025 * 1e-4, update_lr_batch=100000, max_depth=100):
    """options.go:38-58 defaults, wordemb.go:10-18 choices."""
    return W2vCfg(dim, window, 0 if optimizer == "hs" else 1, 0 if model == "skipgram" else 1, neg_samples,
                  init_lr, min_lr, update_lr_batch, max_depth)


def w2v_train_slice(cfg, doc, lo, hi, keep_mask, param, aux, paths, sigtab, lcg, lr, trained_cnt, corpus_len):
    """In-place on param/aux; returns (lr, trained_cnt)."""
    off, nodes, codes = paths
    doc = np.ascontiguousarray(doc, np.int32)
    lr_c = C.c_double(lr)
    cnt_c = C.c_int64(trained_cnt)
    km = np.ascontiguousarray(keep_mask, np.uint8) if keep_mask is not None else None
    lib().orc_w2v_train_slice(C.byref(cfg), _p(doc, C.c_int32), C.c_int64(lo), C.c_int64(hi),
                              _p(km, C.c_uint8) if km is not None else None, _p(param, C.c_double),
                              _p(aux, C.c_double), C.c_int64(param.shape[0]), _p(off, C.c_int64),
                              _p(nodes, C.c_int32), _p(codes, C.c_uint8), _p(sigtab, C.c_double), C.byref(lcg),
                              C.byref(lr_c), C.byref(cnt_c), C.c_int64(corpus_len))
    return lr_c.value, cnt_c.value


def w2v_train_range(cfg, doc, clip_lo, clip_hi, walk_lo, walk_hi, param, aux, paths, sigtab, lcg, lr, trained_cnt, corpus_len):
    """positions [walk_lo, walk_hi) of the slice [clip_lo, clip_hi) (windows clipped at the SLICE's ends): one segment of a
    data-parallel pass of the device library.  In-place on param/aux; returns (lr, trained_cnt)."""
    off, nodes, codes = paths
    doc = np.ascontiguousarray(doc, np.int32)
    lr_c = C.c_double(lr)
    cnt_c = C.c_int64(trained_cnt)
    lib().orc_w2v_train_range(C.byref(cfg), _p(doc, C.c_int32), C.c_int64(clip_lo), C.c_int64(clip_hi), C.c_int64(walk_lo),
                              C.c_int64(walk_hi), None, _p(param, C.c_double), _p(aux, C.c_double), C.c_int64(param.shape[0]),
                              _p(off, C.c_int64), _p(nodes, C.c_int32), _p(codes, C.c_uint8), _p(sigtab, C.c_double), C.byref(lcg),
                              C.byref(lr_c), C.byref(cnt_c), C.c_int64(corpus_len))
    return lr_c.value, cnt_c.value


def w2v_train_hogwild(cfg, doc, threads, keep_mask, param, aux, paths, sigtab, lr, corpus_len):
    off, nodes, codes = paths
    doc = np.ascontiguousarray(doc, np.int32)
    lr_c = C.c_double(lr)
    km = np.ascontiguousarray(keep_mask, np.uint8) if keep_mask is not None else None
    lib().orc_w2v_train_hogwild(C.byref(cfg), _p(doc, C.c_int32), C.c_int64(doc.size), C.c_int(threads),
                                _p(km, C.c_uint8) if km is not None else None, _p(param, C.c_double),
                                _p(aux, C.c_double), C.c_int64(param.shape[0]), _p(off, C.c_int64),
                                _p(nodes, C.c_int32), _p(codes, C.c_uint8), _p(sigtab, C.c_double),
                                C.byref(lr_c), C.c_int64(corpus_len))
    return lr_c.value


# ---------------------------------------------------------------- k-NN search --
def norm64(v):
    v = np.ascontiguousarray(v, np.float64)
    lib().orc_norm64.restype = C.c_double
    return lib().orc_norm64(_p(v, C.c_double), C.c_int(v.size))


def cosine64(v1, v2, n1=None, n2=None):
    v1 = np.ascontiguousarray(v1, np.float64)
    v2 = np.ascontiguousarray(v2, np.float64)
    lib().orc_cosine64.restype = C.c_double
    return lib().orc_cosine64(_p(v1, C.c_double), _p(v2, C.c_double), C.c_int(v1.size),
                              C.c_double(norm64(v1) if n1 is None else n1), C.c_double(norm64(v2) if n2 is None else n2))


def knn_search(items, query, k, ignore=-1, norms=None, qnorm=None):
    """Searcher.Search (search.go:92-134): returns (idx[count], sim[count], rank[count]); idx -1 = empty neighbour."""
    items = np.ascontiguousarray(items, np.float64)
    query = np.ascontiguousarray(query, np.float64)
    V, D = items.shape
    if norms is None:
        norms = np.array([norm64(items[i]) for i in range(V)], np.float64)
    norms = np.ascontiguousarray(norms, np.float64)
    qn = norm64(query) if qnorm is None else qnorm
    idx = np.zeros(max(k, 1), np.int64)
    sim = np.zeros(max(k, 1), np.float64)
    rank = np.zeros(max(k, 1), np.int32)
    lib().orc_knn_search.restype = C.c_int
    n = lib().orc_knn_search(_p(items, C.c_double), _p(norms, C.c_double), C.c_int64(V), C.c_int(D), _p(query, C.c_double),
                             C.c_double(qn), C.c_int(k), C.c_int64(ignore), _p(idx, C.c_int64), _p(sim, C.c_double),
                             _p(rank, C.c_int32))
    return idx[:n], sim[:n], rank[:n]


def knn_search_batch(items, queries, k, norms, ignore=None):
    """Q independent Searcher.Search calls, OpenMP over the queries (set_threads); returns (idx [Q,k], sim [Q,k], count [Q])"""
    items = np.ascontiguousarray(items, np.float64)
    queries = np.ascontiguousarray(queries, np.float64)
    norms = np.ascontiguousarray(norms, np.float64)
    Q, (V, D) = queries.shape[0], items.shape
    idx = np.zeros((Q, k), np.int64); sim = np.zeros((Q, k), np.float64); rank = np.zeros((Q, k), np.int32); cnt = np.zeros(Q, np.int32)
    ig = np.ascontiguousarray(ignore, np.int64) if ignore is not None else None
    lib().orc_knn_search_batch(_p(items, C.c_double), _p(norms, C.c_double), C.c_int64(V), C.c_int(D), _p(queries, C.c_double), C.c_int(Q),
                               C.c_int(k), _p(ig, C.c_int64) if ig is not None else None, _p(idx, C.c_int64), _p(sim, C.c_double),
                               _p(rank, C.c_int32), _p(cnt, C.c_int32))
    return idx, sim, cnt


# ---------------------------------------------------------------- user-behaviour cache / key assembly --
def ubcache_filter(ts, items, max_ts, max_len):
    """TimeSeq.Filter (cache.go:71-94) on a newest-first sequence -> the selected item ids"""
    ts = np.ascontiguousarray(ts, np.int64)
    items = np.ascontiguousarray(items, np.int32)
    out = np.zeros(max(ts.size, 1), np.int32)
    lib().orc_ubcache_filter.restype = C.c_int64
    n = lib().orc_ubcache_filter(_p(ts, C.c_int64), _p(items, C.c_int32), C.c_int64(ts.size), C.c_int64(max_ts),
                                 C.c_int64(max_len), _p(out, C.c_int32))
    return out[:n]


def assemble_keys(off, seq_items, seq_ts, user_table, item_table, users, items, ts, T):
    off = np.ascontiguousarray(off, np.int64)
    seq_items = np.ascontiguousarray(seq_items, np.int32)
    seq_ts = np.ascontiguousarray(seq_ts, np.int64)
    ut = np.ascontiguousarray(user_table, np.float32)
    it = np.ascontiguousarray(item_table, np.float32)
    users = np.ascontiguousarray(users, np.int32)
    items = np.ascontiguousarray(items, np.int32)
    ts = np.ascontiguousarray(ts, np.int64)
    rows = users.size
    ub = np.empty((rows, T), np.int32)
    uf = np.empty((rows, ut.shape[1]), np.float32)
    cf = np.empty((rows, it.shape[1]), np.float32)
    lib().orc_assemble_keys(_p(off, C.c_int64), _p(seq_items, C.c_int32), _p(seq_ts, C.c_int64), C.c_int64(off.size - 1),
                            _p(ut, C.c_float), C.c_int(ut.shape[1]), _p(it, C.c_float), C.c_int64(it.shape[0]),
                            C.c_int(it.shape[1]), _p(users, C.c_int32), _p(items, C.c_int32), _p(ts, C.c_int64),
                            C.c_int64(rows), C.c_int(T), _p(ub, C.c_int32), _p(uf, C.c_float), _p(cf, C.c_float))
    return ub, uf, cf


def batch_predict_rows(emb, off, seq_items, seq_ts, user_table, item_table, users, items, ts, T):
    """recommend.BatchPredict's row assembly (rcmd.go:277-325) in id form: X [n, XCols] float32 and failed [n].
    A key whose user / item has no feature row (GetUserFeature / GetItemFeature error, rcmd.go:474-491) becomes the
    ALL-zero row (rcmd.go:299-302); a failing FIRST key raises (rcmd.go:293-296).  The caller scores X with
    CtrModel.predict (recSys.Predict, rcmd.go:327).  The reference's `err` of the LAST key leaks into the return value
    (named result, rcmd.go:291): callers check failed[-1]."""
    users = np.ascontiguousarray(users, np.int32)
    items = np.ascontiguousarray(items, np.int32)
    ut = np.ascontiguousarray(user_table, np.float32)
    it = np.ascontiguousarray(item_table, np.float32)
    failed = (users < 0) | (users >= ut.shape[0]) | (items < 0) | (items >= it.shape[0])
    if users.size and failed[0]:
        raise ValueError("get sample vector error: first key has no features")
    ub, uf, cf = assemble_keys(off, seq_items, seq_ts, ut, it, users, items, ts, T)
    X = assemble_rows(emb, ub, items, uf, cf)
    X[failed] = 0.0
    return X, failed.astype(np.uint8)


# ---------------------------------------------------------------- corpus / dictionary --
def corpus_build(keys, min_count=5, max_count=-1):
    """(idoc, id2key, cfs, indexed): memory.go:53-102 + dictionary.go:70-81 for integer tokens"""
    keys = np.ascontiguousarray(keys, np.int64)
    n = keys.size
    idoc = np.empty(n, np.int32)
    id2key = np.empty(n, np.int64)
    cfs = np.empty(n, np.int64)
    indexed = np.empty(n, np.int32)
    m = C.c_int64(0)
    lib().orc_corpus_build.restype = C.c_int64
    V = lib().orc_corpus_build(_p(keys, C.c_int64), C.c_int64(n), C.c_int64(min_count), C.c_int64(max_count),
                               _p(idoc, C.c_int32), _p(id2key, C.c_int64), _p(cfs, C.c_int64), _p(indexed, C.c_int32),
                               C.byref(m))
    return idoc, id2key[:V].copy(), cfs[:V].copy(), indexed[:m.value].copy()


def subsample_probs(cfs, threshold=1e-3):
    cfs = np.ascontiguousarray(cfs, np.int64)
    out = np.empty(cfs.size, np.float64)
    lib().orc_subsample_probs(_p(cfs, C.c_int64), C.c_int64(cfs.size), C.c_double(threshold), _p(out, C.c_double))
    return out
