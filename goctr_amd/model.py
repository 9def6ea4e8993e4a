"""Host mirror of go-ctr's ``model`` package for the DIN / YouTube-DNN path.

Reference: model/model.go (Model interface :16-25, Train :27, InitForwardOnlyVm :215, Predict :242),
model/din/din.go (DinNet, NewDinNet :171, NewDinNetFromJson :82, Marshal :62) and
model/youtube/dnn.go (YoutubeDnn, NewYoutubeDnn :119, NewYoutubeDnnFromJson :63, Marshal :49).
The gorgonia graph / VM methods of the Go interface have no meaning for a device model; the handle
returned by ``Vm()`` is the opaque C-ABI model handle.  Everything numeric happens behind
include/goctr.h -- this file only moves buffers and mirrors names, argument order and error behaviour.
"""
from __future__ import annotations

import ctypes as C
import json
import logging

import numpy as np

from . import capi
from .recommend import SampleInfo

log = logging.getLogger("goctr")

# model/din/din.go:14-19
mlp0_1 = 200
mlp1_2 = 80

DIN, YOUTUBE = 0, 1
ATT_COSINE, ATT_EUCLID = 0, 1
_TENSORS = {"mlp0": 0, "mlp1": 1, "mlp2": 2, "att0": 3}


class _CtrNet:
    kind = DIN

    def __init__(self, uProfileDim, uBehaviorSize, uBehaviorDim, iFeatureDim, cFeatureDim, att=ATT_COSINE,
                 d0=0.0, d1=0.0):
        self.uProfileDim, self.uBehaviorSize, self.uBehaviorDim = uProfileDim, uBehaviorSize, uBehaviorDim
        self.iFeatureDim, self.cFeatureDim = iFeatureDim, cFeatureDim
        self.d0, self.d1 = d0, d1
        self.att = att
        self._h = C.c_void_p()
        L = capi.init()
        self.cfg = capi.CtrCfg(self.kind, att, uProfileDim, uBehaviorSize, uBehaviorDim, cFeatureDim, mlp0_1, mlp1_2)
        capi.check(L.goctr_model_create(C.byref(self.cfg), C.byref(self._h)))
        self.predBatchSize = None

    # --- Model interface (model.go:16-25) ---------------------------------------------------
    def Learnable(self):
        return ["mlp0", "mlp1", "mlp2"] + (["att0"] if self.kind == DIN else [])

    def Vm(self):
        return self._h

    def SetVM(self, vm):  # kept for surface parity; the device model is its own "VM"
        pass

    @property
    def I(self):
        return self.uProfileDim + self.uBehaviorDim + self.iFeatureDim + self.cFeatureDim

    def _shape(self, name):
        return {"mlp0": (self.I, mlp0_1), "mlp1": (mlp0_1, mlp1_2), "mlp2": (mlp1_2, 1),
                "att0": (1, self.uBehaviorSize)}[name]

    def set_weights(self, name, arr):
        a = capi.f32(arr).ravel()
        capi.check(capi.load().goctr_model_set_weights(self._h, C.c_int(_TENSORS[name]), capi.ptr(a, C.c_float),
                                                       C.c_size_t(a.size)))

    def get_weights(self, name):
        shp = self._shape(name)
        a = np.empty(int(np.prod(shp)), np.float32)
        capi.check(capi.load().goctr_model_get_weights(self._h, C.c_int(_TENSORS[name]), capi.ptr(a, C.c_float),
                                                       C.c_size_t(a.size)))
        return a.reshape(shp)

    # --- optimizer state (checkpoint / resume) -----------------------------------
    def get_moments(self, name, which):
        shp = self._shape(name)
        a = np.empty(int(np.prod(shp)), np.float32)
        capi.check(capi.load().goctr_model_get_moments(self._h, C.c_int(_TENSORS[name]), C.c_int(which),
                                                       capi.ptr(a, C.c_float), C.c_size_t(a.size)))
        return a.reshape(shp)

    def set_moments(self, name, which, arr):
        a = capi.f32(arr).ravel()
        capi.check(capi.load().goctr_model_set_moments(self._h, C.c_int(_TENSORS[name]), C.c_int(which),
                                                       capi.ptr(a, C.c_float), C.c_size_t(a.size)))

    @property
    def step(self):
        v = C.c_uint32()
        capi.check(capi.load().goctr_model_get_step(self._h, C.byref(v)))
        return int(v.value)

    @step.setter
    def step(self, v):
        capi.check(capi.load().goctr_model_set_step(self._h, C.c_uint32(int(v))))

    def set_embedding_training(self, lr):
        """EXTENSION (no reference counterpart: go-ctr trains with frozen embeddings): lr > 0 lets the following training
        steps on id-mode datasets also update the EmbeddingTable rows by SGD scatter-add; 0 switches it off."""
        capi.check(capi.load().goctr_model_set_embedding_training(self._h, C.c_double(lr)))
        return self

    def emb_plan(self):
        """the resident sparse plan of the last embedding-training call as a dict of numpy arrays (tests / tools)"""
        L = capi.load()
        nb, npair, nslot = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        capi.check(L.goctr_model_get_emb_plan(self._h, C.byref(nb), C.byref(npair), C.byref(nslot), None, None, None, None, None, None, None))
        p = {"pair": np.empty(npair.value, np.int32), "pslot": np.empty(npair.value, np.int32), "pid": np.empty(npair.value, np.int32),
             "slot_id": np.empty(nslot.value, np.int32), "slot_off": np.empty(nslot.value + nb.value, np.uint32),
             "pair_off": np.empty(nb.value + 1, np.int64), "slot_base": np.empty(nb.value + 1, np.int64)}
        capi.check(L.goctr_model_get_emb_plan(self._h, None, None, None, capi.ptr(p["pair"], C.c_int32), capi.ptr(p["pslot"], C.c_int32),
                                              capi.ptr(p["pid"], C.c_int32), capi.ptr(p["slot_id"], C.c_int32),
                                              capi.ptr(p["slot_off"], C.c_uint32), capi.ptr(p["pair_off"], C.c_int64),
                                              capi.ptr(p["slot_base"], C.c_int64)))
        return p

    def emb_plan_build_ms(self):
        """(wall milliseconds, batches) of the resident plan's build"""
        ms, nb = C.c_double(0), C.c_int64(0)
        capi.check(capi.load().goctr_model_emb_plan_build_ms(self._h, C.byref(ms), C.byref(nb)))
        return ms.value, nb.value

    def sparse_exchange_bytes(self):
        """bytes this rank sent in the last step's sparse-gradient exchange (0 without a communicator)"""
        v = C.c_double(0)
        capi.check(capi.load().goctr_model_sparse_exchange_bytes(self._h, C.byref(v)))
        return v.value

    def replica(self, rank):
        """the replica a multi-device training call (cfg.devices = n) keeps on engine `rank`, as a borrowed model object
        (rank 0: this model); None before the first such call"""
        h = C.c_void_p()
        capi.check(capi.load().goctr_model_replica(self._h, C.c_int(rank), C.byref(h)))
        if not h:
            return None
        if rank == 0:
            return self
        r = object.__new__(type(self))
        r.__dict__.update(self.__dict__)
        r._h, r._borrowed = h, True
        return r

    def init_gaussian(self, rng):
        """G.Gaussian(0, 1) weights, att0 = 1 (din.go:181-191; dnn.go:125-127)."""
        for n in ("mlp0", "mlp1", "mlp2"):
            self.set_weights(n, rng.standard_normal(self._shape(n)).astype(np.float32))
        if self.kind == DIN:
            self.set_weights("att0", np.ones(self._shape("att0"), np.float32))
        return self

    def Marshal(self, optimizer=False) -> bytes:
        """din.go:62-80 / dnn.go:49-61: the dinModel / mlpModel JSON layout.  optimizer=True adds the keys a resume
        needs and the reference's JSON lacks ("adamStep", "<tensor>_m", "<tensor>_v"); Go's json.Unmarshal ignores
        unknown keys, so the file stays loadable by NewDinNetFromJson / NewYoutubeDnnFromJson there."""
        d = {"uProfileDim": self.uProfileDim, "uBehaviorSize": self.uBehaviorSize, "uBehaviorDim": self.uBehaviorDim,
             "iFeatureDim": self.iFeatureDim, "cFeatureDim": self.cFeatureDim}
        for n in self.Learnable():
            d[n] = [float(x) for x in self.get_weights(n).ravel()]
        if optimizer:
            d["adamStep"] = self.step
            for n in self.Learnable():
                d[n + "_m"] = [float(x) for x in self.get_moments(n, 0).ravel()]
                d[n + "_v"] = [float(x) for x in self.get_moments(n, 1).ravel()]
        return json.dumps(d).encode()

    def close(self):
        if self._h and not getattr(self, "_borrowed", False):
            capi.load().goctr_model_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DinNet(_CtrNet):
    kind = DIN


class YoutubeDnn(_CtrNet):
    kind = YOUTUBE

    def __init__(self, *a, **kw):
        kw.pop("att", None)
        super().__init__(*a, **kw)


def NewDinNet(uProfileDim, uBehaviorSize, uBehaviorDim, iFeatureDim, cFeatureDim, rng=None, att=ATT_COSINE):
    """din.go:171-211.  Fatal (here: ValueError) when uBehaviorDim != iFeatureDim (din.go:176-178)."""
    if uBehaviorDim != iFeatureDim:
        raise ValueError(f"uBehaviorDim {uBehaviorDim} != iFeatureDim {iFeatureDim}")
    m = DinNet(uProfileDim, uBehaviorSize, uBehaviorDim, iFeatureDim, cFeatureDim, att=att, d0=0.005, d1=0.005)
    return m.init_gaussian(rng or np.random.default_rng())


def NewYoutubeDnn(uProfileDim, uBehaviorSize, uBehaviorDim, iFeatureDim, cFeatureDim, rng=None):
    """dnn.go:119-142"""
    m = YoutubeDnn(uProfileDim, uBehaviorSize, uBehaviorDim, iFeatureDim, cFeatureDim, d0=0.003, d1=0.003)
    return m.init_gaussian(rng or np.random.default_rng())


def _from_json(cls, data):
    d = json.loads(data)
    m = cls(d["uProfileDim"], d["uBehaviorSize"], d["uBehaviorDim"], d["iFeatureDim"], d["cFeatureDim"])
    for n in m.Learnable():
        m.set_weights(n, np.asarray(d[n], np.float32))
    if "adamStep" in d:  # written by Marshal(optimizer=True): resume Adam where it stopped
        for n in m.Learnable():
            m.set_moments(n, 0, np.asarray(d[n + "_m"], np.float32))
            m.set_moments(n, 1, np.asarray(d[n + "_v"], np.float32))
        m.step = d["adamStep"]
    return m  # d0 = d1 = 0 like the Go constructors-from-JSON (din.go:136-147): no dropout at predict


def NewDinNetFromJson(data: bytes) -> DinNet:
    """din.go:82-147"""
    return _from_json(DinNet, data)


def NewYoutubeDnnFromJson(data: bytes) -> YoutubeDnn:
    """dnn.go:63-109"""
    return _from_json(YoutubeDnn, data)


def Train(uProfileDim, uBehaviorSize, uBehaviorDim, iFeatureDim, cFeatureDim, numExamples, batchSize, epochs,
          earlyStop, si: SampleInfo, inputs: np.ndarray, targets: np.ndarray, m: _CtrNet, dropout_seed=42, devices=0):
    """model.go:27-213.  ``inputs`` [numExamples, XCols] float32, ``targets`` [numExamples(,1)].
    Returns the per-epoch costs (the Go version only logs them: model.go:205).
    Like the reference, training applies Dropout(m.d0) / Dropout(m.d1) whenever the model carries non-zero rates
    (NewDinNet: 0.005, din.go:204-205,307-312; NewYoutubeDnn: 0.003): a counter-hash mask stream on the device, seeded
    by ``dropout_seed``.  The reference draws its masks from Go's math/rand, which cannot be reproduced without Go, so
    the mask BITS are unpinned; the distribution is the same.  ``dropout_seed=None`` is the explicit opt-out.
    ``devices=n`` (after ``capi.init_devices``): the same single call, data-parallel over n engines (batchSize stays the
    global batch; no reference counterpart)."""
    X = capi.f32(inputs)
    Y = capi.f32(targets).ravel()
    if X.shape[0] != numExamples or Y.shape[0] != numExamples:
        raise ValueError("numExamples does not match inputs/targets")
    cfg = capi.default_train_cfg(batch=batchSize, epochs=epochs, early_stop=earlyStop, dropout_mode=0, devices=devices)
    if dropout_seed is not None and (m.d0 > 0 or m.d1 > 0):
        cfg.dropout_mode, cfg.p0, cfg.p1, cfg.seed = 2, m.d0, m.d1, dropout_seed
    costs = np.zeros(max(epochs, 1), np.float32)
    ran = C.c_int(0)
    r = si.as_ranges()
    capi.check(capi.load().goctr_train_dense(m._h, capi.ptr(X, C.c_float), capi.ptr(Y, C.c_float),
                                             C.c_int64(numExamples), C.c_int(X.shape[1]), capi.ptr(r, C.c_int),
                                             C.byref(cfg), capi.ptr(costs, C.c_float), C.byref(ran)))
    for i in range(ran.value):
        log.info("Epoch %d | cost %v".replace("%v", "%s"), i, costs[i])
    return costs[:ran.value]


def InitForwardOnlyVm(uProfileDim, uBehaviorSize, uBehaviorDim, iFeatureDim, cFeatureDim, batchSize, m: _CtrNet):
    """model.go:215-240: fixes the predict batch size (the device model needs no separate graph)."""
    if (uProfileDim, uBehaviorSize, uBehaviorDim, cFeatureDim) != (m.uProfileDim, m.uBehaviorSize, m.uBehaviorDim,
                                                                    m.cFeatureDim):
        raise ValueError("dims do not match the model")
    m.predBatchSize = batchSize


def Predict(m: _CtrNet, numExamples, batchSize, si: SampleInfo, inputs: np.ndarray) -> np.ndarray:
    """model.go:242-352: returns y [numExamples] float32."""
    X = capi.f32(inputs)
    y = np.empty(numExamples, np.float32)
    r = si.as_ranges()
    capi.check(capi.load().goctr_predict_dense(m._h, capi.ptr(X, C.c_float), C.c_int64(numExamples),
                                               C.c_int(X.shape[1]), capi.ptr(r, C.c_int), C.c_int(batchSize),
                                               capi.ptr(y, C.c_float)))
    return y


def loss_grad(m: _CtrNet, si: SampleInfo, X, Y, B=None, cfg=None, step=0, m0=None, m1=None):
    """parity entry (goctr_loss_grad_dense): cost, grads dict, y for ONE batch, no update."""
    X = capi.f32(X)
    Y = capi.f32(Y).ravel()
    valid = X.shape[0]
    B = B or valid
    cfg = cfg or capi.default_train_cfg(batch=B, dropout_mode=0)
    g = {n: np.zeros(m._shape(n), np.float32) for n in ("mlp0", "mlp1", "mlp2", "att0")}
    y = np.zeros(B, np.float32)
    cost = C.c_float(0)
    r = si.as_ranges()
    m0 = capi.f32(m0) if m0 is not None else None
    m1 = capi.f32(m1) if m1 is not None else None
    capi.check(capi.load().goctr_loss_grad_dense(
        m._h, capi.ptr(X, C.c_float), capi.ptr(Y, C.c_float), C.c_int(valid), C.c_int(B), C.c_int(X.shape[1]),
        capi.ptr(r, C.c_int), C.byref(cfg), C.c_uint32(step), capi.ptr(m0, C.c_float), capi.ptr(m1, C.c_float),
        C.byref(cost), capi.ptr(g["mlp0"], C.c_float), capi.ptr(g["mlp1"], C.c_float), capi.ptr(g["mlp2"], C.c_float),
        capi.ptr(g["att0"], C.c_float), capi.ptr(y, C.c_float)))
    return cost.value, g, y


# ------------------------------------------------------------------- id (performance) mode
class EmbeddingTable:
    """item-embedding table [V, D] float32 resident in HBM (replaces itemEmbeddingMap, rcmd.go:31-32)."""

    def __init__(self, rows: np.ndarray):
        rows = capi.f32(rows)
        self.V, self.D = rows.shape
        self._h = C.c_void_p()
        capi.init()
        capi.check(capi.load().goctr_emb_create(C.c_int64(self.V), C.c_int(self.D), capi.ptr(rows, C.c_float),
                                                C.byref(self._h)))

    def replica(self, rank):
        """this table's replica on engine `rank` (multi-device training), borrowed; None before the first such call"""
        h = C.c_void_p()
        capi.check(capi.load().goctr_emb_replica(self._h, C.c_int(rank), C.byref(h)))
        if not h:
            return None
        if rank == 0:
            return self
        r = object.__new__(EmbeddingTable)
        r.V, r.D, r._h, r._borrowed = self.V, self.D, h, True
        return r

    def get_rows(self, first=0, n=None):
        n = self.V - first if n is None else n
        out = np.empty((n, self.D), np.float32)
        capi.check(capi.load().goctr_emb_get_rows(self._h, C.c_int64(first), C.c_int64(n), capi.ptr(out, C.c_float)))
        return out

    def gather_rows(self, ub_ids, item_ids, user_feat, ctx_feat):
        """GetSampleVector's row assembly (rcmd.go:497-533) on the device; bit-exact copies."""
        ub_ids, item_ids = capi.i32(ub_ids), capi.i32(item_ids)
        uf, cf = capi.f32(user_feat), capi.f32(ctx_feat)
        rows, T = ub_ids.shape
        X = np.empty((rows, uf.shape[1] + T * self.D + self.D + cf.shape[1]), np.float32)
        capi.check(capi.load().goctr_gather_rows(self._h, capi.ptr(ub_ids, C.c_int32), capi.ptr(item_ids, C.c_int32),
                                                 capi.ptr(uf, C.c_float), C.c_int(uf.shape[1]), capi.ptr(cf, C.c_float),
                                                 C.c_int(cf.shape[1]), C.c_int(T), C.c_int64(rows),
                                                 capi.ptr(X, C.c_float)))
        return X

    def close(self):
        if self._h and not getattr(self, "_borrowed", False):
            capi.load().goctr_emb_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Dataset:
    """sample rows resident in HBM: dense TrainSample rows or (ids + side features)."""

    def __init__(self, handle, rows):
        self._h, self.rows = handle, rows

    @staticmethod
    def dense(X, Y, si: SampleInfo):
        X = capi.f32(X)
        Yp = capi.f32(Y).ravel() if Y is not None else None
        h = C.c_void_p()
        r = si.as_ranges()
        capi.init()
        capi.check(capi.load().goctr_dataset_create_dense(capi.ptr(X, C.c_float), capi.ptr(Yp, C.c_float),
                                                          C.c_int64(X.shape[0]), C.c_int(X.shape[1]),
                                                          capi.ptr(r, C.c_int), C.byref(h)))
        return Dataset(h, X.shape[0])

    @staticmethod
    def ids(ub_ids, item_ids, user_feat, ctx_feat, Y):
        ub_ids, item_ids = capi.i32(ub_ids), capi.i32(item_ids)
        uf, cf = capi.f32(user_feat), capi.f32(ctx_feat)
        Yp = capi.f32(Y).ravel() if Y is not None else None
        rows, T = ub_ids.shape
        h = C.c_void_p()
        capi.init()
        capi.check(capi.load().goctr_dataset_create_ids(capi.ptr(ub_ids, C.c_int32), capi.ptr(item_ids, C.c_int32),
                                                        capi.ptr(uf, C.c_float), C.c_int(uf.shape[1]),
                                                        capi.ptr(cf, C.c_float), C.c_int(cf.shape[1]), C.c_int(T),
                                                        capi.ptr(Yp, C.c_float), C.c_int64(rows), C.byref(h)))
        return Dataset(h, rows)

    @staticmethod
    def keys(ubc, user_table, item_table, users, items, ts, Y, T):
        """device-side sample assembly (replaces GetSample / GetSampleVector's per-sample host gather, rcmd.go:339-536):
        ubc = ubcache.UserBehaviorCache; users / items = DENSE row indices into user_table / item_table (and into
        ubc.user_index() order); ts = the samples' timestamps"""
        users = np.ascontiguousarray(users, np.int32)
        items = np.ascontiguousarray(items, np.int32)
        ts = np.ascontiguousarray(ts, np.int64)
        ut = capi.f32(user_table)
        it = capi.f32(item_table)
        y = None if Y is None else capi.f32(Y)
        h = C.c_void_p()
        capi.check(capi.load().goctr_dataset_create_keys(ubc.device(), capi.ptr(ut, C.c_float), C.c_int64(ut.shape[0]),
                                                         C.c_int(ut.shape[1]), capi.ptr(it, C.c_float), C.c_int64(it.shape[0]),
                                                         C.c_int(it.shape[1]), capi.ptr(users, C.c_int32),
                                                         capi.ptr(items, C.c_int32), capi.ptr(ts, C.c_int64),
                                                         capi.ptr(y, C.c_float), C.c_int64(users.size), C.c_int(T), C.byref(h)))
        d = Dataset(h, users.size)
        d._dims = (T, ut.shape[1], it.shape[1])
        return d

    def get_ids(self):
        """(ub_ids [rows,T], user_feat [rows,U], ctx_feat [rows,C]) of a dataset built by keys()"""
        T, U, Cc = self._dims
        ub = np.empty((self.rows, T), np.int32)
        uf = np.empty((self.rows, U), np.float32)
        cf = np.empty((self.rows, Cc), np.float32)
        capi.check(capi.load().goctr_dataset_get_ids(self._h, capi.ptr(ub, C.c_int32), capi.ptr(uf, C.c_float),
                                                     capi.ptr(cf, C.c_float)))
        return ub, uf, cf

    def close(self):
        if self._h:
            capi.load().goctr_dataset_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def train_dataset(m: _CtrNet, ds: Dataset, cfg, emb: EmbeddingTable | None = None):
    costs = np.zeros(max(cfg.epochs, 1), np.float32)
    ran = C.c_int(0)
    capi.check(capi.load().goctr_train_dataset(m._h, emb._h if emb else None, ds._h, C.byref(cfg),
                                               capi.ptr(costs, C.c_float), C.byref(ran)))
    return costs[:ran.value]


def train_steps(m: _CtrNet, ds: Dataset, cfg, n_steps, first_batch=0, emb: EmbeddingTable | None = None,
                want_costs=False):
    costs = np.zeros(max(n_steps, 1), np.float32) if want_costs else None
    capi.check(capi.load().goctr_train_steps(m._h, emb._h if emb else None, ds._h, C.byref(cfg),
                                             C.c_int64(first_batch), C.c_int(n_steps), capi.ptr(costs, C.c_float)))
    return costs[:n_steps] if want_costs else None


def predict_dataset(m: _CtrNet, ds: Dataset, batch, emb: EmbeddingTable | None = None):
    y = np.empty(ds.rows, np.float32)
    capi.check(capi.load().goctr_predict_dataset(m._h, emb._h if emb else None, ds._h, C.c_int(batch),
                                                 capi.ptr(y, C.c_float)))
    return y


def predict_steps(m: _CtrNet, ds: Dataset, batch, n_batches, first_batch=0, emb: EmbeddingTable | None = None):
    capi.check(capi.load().goctr_predict_steps(m._h, emb._h if emb else None, ds._h, C.c_int(batch),
                                               C.c_int64(first_batch), C.c_int(n_batches)))
