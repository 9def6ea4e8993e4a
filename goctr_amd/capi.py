"""ctypes binding of the C-ABI in include/goctr.h (libgoctr_hip.so).

This is the Python stand-in for the cgo stub shown in INTEGRATION.md: the same entry points, the
same argument meaning.  There is no CPU fallback -- if the shared library or a HIP device is
missing every call raises ``GoctrError``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GOCTR_LIB: load another build of the same C-ABI (tests/test_bench_dryrun.py drives bench.py --gpus 8 against a stub that
# exports every symbol of include/goctr.h and computes nothing -- launcher / rendezvous / JSON plumbing on a GPU-less box)
LIB_PATH = os.environ.get("GOCTR_LIB") or os.path.join(_HERE, "libgoctr_hip.so")


class GoctrError(RuntimeError):
    pass


class CtrCfg(C.Structure):
    _fields_ = [("kind", C.c_int), ("att", C.c_int), ("U", C.c_int), ("T", C.c_int), ("D", C.c_int),
                ("C", C.c_int), ("H1", C.c_int), ("H2", C.c_int)]


class TrainCfg(C.Structure):
    _fields_ = [("batch", C.c_int), ("epochs", C.c_int), ("early_stop", C.c_int), ("lr", C.c_double),
                ("l2", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("adam_div_by_batch", C.c_int), ("adam_l2_before_batch_div", C.c_int), ("dropout_mode", C.c_int),
                ("p0", C.c_float), ("p1", C.c_float), ("seed", C.c_uint32), ("devices", C.c_int)]


class MlpCfg(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("units", C.c_int * 8), ("activation", C.c_int), ("solver", C.c_int),
                ("alpha", C.c_double), ("lr_init", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_double), ("momentum", C.c_double), ("nesterov", C.c_int), ("batch_normalize", C.c_int),
                ("weight_decay", C.c_double), ("batch", C.c_int), ("max_iter", C.c_int),
                ("n_iter_no_change", C.c_int), ("tol", C.c_double)]


class W2vCfg(C.Structure):
    _fields_ = [("dim", C.c_int), ("window", C.c_int), ("optimizer", C.c_int), ("model", C.c_int),
                ("neg_samples", C.c_int), ("init_lr", C.c_double), ("min_lr", C.c_double),
                ("update_lr_batch", C.c_int64), ("max_depth", C.c_int), ("deterministic", C.c_int),
                ("streams", C.c_int), ("slices", C.c_int), ("devices", C.c_int), ("exchange_every", C.c_int64)]


# every symbol include/goctr.h declares (tests/test_capi_symbols.py checks the list against the header)
SYMBOLS = [
    "goctr_init", "goctr_init_devices", "goctr_engine_count", "goctr_engine_call_ms", "goctr_engine_select", "goctr_comm_group_enable", "goctr_device_count", "goctr_sync", "goctr_last_error", "goctr_version", "goctr_device_info",
    "goctr_comm_unique_id", "goctr_comm_init", "goctr_comm_world", "goctr_comm_capture_mode", "goctr_comm_allreduce_f64", "goctr_comm_destroy",
    "goctr_model_replica", "goctr_emb_replica", "goctr_model_create", "goctr_model_destroy", "goctr_model_set_weights", "goctr_model_get_weights",
    "goctr_model_reset_optimizer", "goctr_model_get_moments", "goctr_model_set_moments", "goctr_model_get_step",
    "goctr_model_set_step", "goctr_model_get_emb_plan", "goctr_model_emb_plan_build_ms", "goctr_model_set_embedding_training", "goctr_model_sparse_exchange_bytes", "goctr_emb_get_rows", "goctr_train_cfg_default", "goctr_train_dense", "goctr_predict_dense",
    "goctr_loss_grad_dense", "goctr_emb_create", "goctr_emb_set_rows", "goctr_emb_destroy", "goctr_gather_rows",
    "goctr_dataset_create_dense", "goctr_dataset_create_ids", "goctr_dataset_destroy", "goctr_train_dataset",
    "goctr_train_steps", "goctr_predict_dataset", "goctr_predict_steps", "goctr_prof_enable", "goctr_prof_reset",
    "goctr_prof_get", "goctr_prof_name", "goctr_prof_kernel", "goctr_mlp_cfg_default", "goctr_mlp_create", "goctr_mlp_destroy",
    "goctr_mlp_nparams", "goctr_mlp_set_params", "goctr_mlp_get_params", "goctr_mlp_loss_grad", "goctr_mlp_fit", "goctr_mlp_fit_resident",
    "goctr_mlp_upload", "goctr_mlp_train_steps", "goctr_mlp_predict", "goctr_w2v_cfg_default", "goctr_w2v_create",
    "goctr_w2v_destroy", "goctr_w2v_set_param", "goctr_w2v_set_aux", "goctr_w2v_get_param", "goctr_w2v_get_aux",
    "goctr_w2v_get_paths", "goctr_huffman_build", "goctr_w2v_train", "goctr_w2v_upload_doc", "goctr_w2v_shard_cuts", "goctr_w2v_train_resident",
    "goctr_w2v_export_f32", "goctr_searcher_create", "goctr_searcher_destroy", "goctr_searcher_search",
    "goctr_ubcache_create", "goctr_ubcache_destroy", "goctr_ubcache_get", "goctr_dataset_create_keys", "goctr_dataset_get_ids",
    "goctr_recsys_create", "goctr_recsys_destroy", "goctr_batch_predict", "goctr_rank",
    "goctr_corpus_create", "goctr_corpus_destroy", "goctr_corpus_append", "goctr_corpus_build", "goctr_corpus_info",
    "goctr_corpus_get_dictionary", "goctr_corpus_get_doc", "goctr_w2v_create_from_corpus", "goctr_w2v_use_corpus",
    "goctr_w2v_get_keep_mask",
]

_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree libgoctr_hip.so (built by __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GoctrError(f"{LIB_PATH} is missing -- run `python -c 'import __graft_entry__ as g; g.build()'`; "
                             "there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.goctr_last_error.restype = C.c_char_p
        _lib.goctr_version.restype = C.c_char_p
        _lib.goctr_prof_name.restype = C.c_char_p
        _lib.goctr_prof_kernel.restype = C.c_char_p
        _lib.goctr_mlp_nparams.restype = C.c_size_t
        for name in ("goctr_model_destroy", "goctr_emb_destroy", "goctr_dataset_destroy", "goctr_mlp_destroy",
                     "goctr_w2v_destroy", "goctr_searcher_destroy", "goctr_ubcache_destroy", "goctr_recsys_destroy", "goctr_train_cfg_default", "goctr_mlp_cfg_default",
                     "goctr_w2v_cfg_default"):
            getattr(_lib, name).restype = None
    return _lib


def check(rc: int):
    if rc != 0:
        raise GoctrError(load().goctr_last_error().decode())


def ptr(a, ty):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ty))


def f32(a):
    return np.ascontiguousarray(a, np.float32)


def i32(a):
    return np.ascontiguousarray(a, np.int32)


_inited = False


def init(device: int | None = None):
    """goctr_init on LOCAL_RANK (one process per GPU)."""
    global _inited
    L = load()
    if device is None:
        if _inited:                 # (already bound -- possibly by init_devices)
            return L
        device = int(os.environ.get("LOCAL_RANK", "0"))
    check(L.goctr_init(C.c_int(device)))
    _inited = True
    return L


def init_devices(device_ids):
    """goctr_init_devices: ONE process drives len(device_ids) ranks (engine k on HIP device device_ids[k]); a repeated device
    id gives several logical ranks on that device joined by the loop-back communicator.  Training calls with
    ``cfg.devices = len(device_ids)`` then run data-parallel inside one call."""
    global _inited
    L = load()
    ids = (C.c_int * len(device_ids))(*[int(x) for x in device_ids])
    check(L.goctr_init_devices(C.c_int(len(device_ids)), ids))
    _inited = True
    return L


def engine_count() -> int:
    n = C.c_int(0)
    check(load().goctr_engine_count(C.byref(n)))
    return n.value


def engine_call_ms(k: int) -> float:
    """device time rank k spent in its part of the last multi-device training call (goctr_engine_call_ms)"""
    ms = C.c_double(0.0)
    check(load().goctr_engine_call_ms(C.c_int(k), C.byref(ms)))
    return ms.value


def engine_select(k: int):
    """bind the calling thread to engine k for the handles it creates (tools / tests)"""
    check(load().goctr_engine_select(C.c_int(k)))


def comm_group_enable(on: bool = True):
    check(load().goctr_comm_group_enable(C.c_int(1 if on else 0)))


def device_count() -> int:
    n = C.c_int(0)
    check(load().goctr_device_count(C.byref(n)))
    return n.value


def sync():
    check(load().goctr_sync())


def device_info():
    name = C.create_string_buffer(256)
    cus = C.c_int(0)
    hbm = C.c_int64(0)
    check(load().goctr_device_info(name, C.c_size_t(256), C.byref(cus), C.byref(hbm)))
    return name.value.decode(), cus.value, hbm.value


def default_train_cfg(**kw) -> TrainCfg:
    c = TrainCfg()
    load().goctr_train_cfg_default(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


PROF_COUNT = 18          # GOCTR_K_COUNT (include/goctr.h)


def prof_enable(on: bool):
    check(load().goctr_prof_enable(C.c_int(1 if on else 0)))


def prof_reset():
    check(load().goctr_prof_reset())


def prof_kernels():
    """{kernel family: symbol of the kernel its last profiled launch ran} (goctr_prof_kernel)"""
    L = load()
    return {L.goctr_prof_name(C.c_int(k)).decode(): L.goctr_prof_kernel(C.c_int(k)).decode() for k in range(PROF_COUNT)}


def prof_get():
    """{kernel family: (total_ms, launches)} since the last reset."""
    L = load()
    out = {}
    for k in range(PROF_COUNT):
        ms = C.c_double(0)
        n = C.c_int64(0)
        check(L.goctr_prof_get(C.c_int(k), C.byref(ms), C.byref(n)))
        out[L.goctr_prof_name(C.c_int(k)).decode()] = (ms.value, n.value)
    return out
