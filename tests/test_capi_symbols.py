"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/goctr.h declares, the ctypes structs match the header layouts, and without a GPU every compute
entry point fails LOUDLY (there is no CPU fallback in the product path)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "goctr.h")


def header_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(goctr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from goctr_amd import capi
    lib = capi.load()
    declared = header_functions()
    assert len(declared) >= 55
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(capi.SYMBOLS) == declared            # the binding's list is the header's list


def test_nm_lists_only_c_linkage_for_the_abi():
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "goctr_amd", "libgoctr_hip.so")],
                         capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(header_functions()) <= exported


def test_struct_layouts_match_header():
    """compile a tiny C program against include/goctr.h and compare sizeof / offsetof with ctypes"""
    from goctr_amd import capi
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "goctr.h"
int main(void) {
  printf("%zu %zu %zu %zu\n", sizeof(goctr_ctr_cfg), sizeof(goctr_train_cfg), sizeof(goctr_mlp_cfg), sizeof(goctr_w2v_cfg));
  printf("%zu %zu %zu\n", offsetof(goctr_train_cfg, lr), offsetof(goctr_train_cfg, dropout_mode), offsetof(goctr_train_cfg, seed));
  printf("%zu %zu %zu\n", offsetof(goctr_mlp_cfg, alpha), offsetof(goctr_mlp_cfg, batch), offsetof(goctr_mlp_cfg, tol));
  printf("%zu %zu\n", offsetof(goctr_w2v_cfg, init_lr), offsetof(goctr_w2v_cfg, deterministic));
  return 0;
}'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    got = list(map(int, out))
    exp = [C.sizeof(capi.CtrCfg), C.sizeof(capi.TrainCfg), C.sizeof(capi.MlpCfg), C.sizeof(capi.W2vCfg),
           capi.TrainCfg.lr.offset, capi.TrainCfg.dropout_mode.offset, capi.TrainCfg.seed.offset,
           capi.MlpCfg.alpha.offset, capi.MlpCfg.batch.offset, capi.MlpCfg.tol.offset,
           capi.W2vCfg.init_lr.offset, capi.W2vCfg.deterministic.offset]
    assert got == exp


def test_defaults_are_the_reference_literals():
    from goctr_amd import capi
    L = capi.load()
    t = capi.default_train_cfg()
    assert (t.lr, t.l2, t.beta1, t.beta2, t.eps) == (0.01, 0.0001, 0.9, 0.999, 1e-8)       # model.go:88
    assert (t.adam_div_by_batch, t.adam_l2_before_batch_div) == (1, 1)
    m = capi.MlpCfg(); L.goctr_mlp_cfg_default(C.byref(m))
    assert (m.alpha, m.lr_init, m.beta1, m.beta2, m.eps, m.momentum, m.nesterov) == (1e-4, 1e-3, .9, .999, 1e-8, .9, 1)
    assert (m.batch, m.max_iter, m.n_iter_no_change, m.tol) == (200, 200, 10, 1e-4)          # basemlp64.go:228-254
    w = capi.W2vCfg(); L.goctr_w2v_cfg_default(C.byref(w))
    assert (w.dim, w.window, w.optimizer, w.model, w.neg_samples, w.max_depth) == (16, 5, 0, 0, 5, 100)
    assert w.init_lr == 0.025 and w.min_lr == 0.025 * 1.0e-4 and w.update_lr_batch == 100000  # options.go:38-58


def _no_gpu():
    from goctr_amd import capi
    return capi.device_count() == 0


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "goctr_amd", "libgoctr_hip.so")), reason="library not built")
def test_no_silent_cpu_fallback():
    """on a box without a GPU the product must fail loudly, never compute on the host"""
    from goctr_amd import capi, model as gm
    if not _no_gpu():
        pytest.skip("GPU present")
    with pytest.raises(capi.GoctrError, match="no HIP device|no CPU fallback"):
        capi.init()
    L = capi.load()
    h = C.c_void_p()
    cfg = capi.CtrCfg(0, 0, 5, 3, 7, 5, 200, 80)
    assert L.goctr_model_create(C.byref(cfg), C.byref(h)) != 0
    assert b"goctr_init" in L.goctr_last_error()
    with pytest.raises(capi.GoctrError):
        gm.DinNet(5, 3, 7, 7, 5)
    x = np.zeros((4, 40), np.float32)
    assert L.goctr_predict_dense(None, capi.ptr(x, C.c_float), C.c_int64(4), C.c_int(40), None, C.c_int(2), None) != 0


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under goctr_amd/ (Python or HIP) may reference it"""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "goctr_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|pyoracle|libgoctr_oracle|goctr_oracle\.h|orc_[a-z0-9_]+\s*\(", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_host_mirror_sample_info():
    from goctr_amd.recommend import ItemEmbDim, SampleInfo, UserBehaviorLen
    si = SampleInfo.from_dims(52, UserBehaviorLen, ItemEmbDim, 53)         # MovieLens widths
    assert si.as_ranges().tolist() == [0, 52, 52, 212, 212, 228, 228, 281]
    assert si.CtxFeatureRange[1] == 281


def _build_cpp_example(tmpdir):
    exe = os.path.join(tmpdir, "example_main")
    subprocess.run(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "goctr_amd", "host", "example_main.cpp"), "-o", exe,
                    "-L" + os.path.join(ROOT, "goctr_amd"), "-lgoctr_hip",
                    "-Wl,-rpath," + os.path.join(ROOT, "goctr_amd")], check=True)
    return exe


def test_cpp_host_mirror_links_and_fails_loudly_without_gpu(tmp_path):
    """goctr_amd/host/goctr.hpp (the C++ twin of the Go surface) compiles against include/goctr.h and links
    the C-ABI; on a GPU-less box the example exits non-zero with the no-device error, it does not compute"""
    exe = _build_cpp_example(str(tmp_path))
    if not _no_gpu():
        pytest.skip("GPU present (covered by the gpu test)")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_cpp_host_mirror_runs_on_gpu(tmp_path):
    exe = _build_cpp_example(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "epochs 5" in r.stdout and "n 118" in r.stdout


def test_profiler_family_count_matches_the_header():
    """capi.PROF_COUNT mirrors the GOCTR_K_* enum of include/goctr.h (a family added there must be visible to bench.py)"""
    import re
    from goctr_amd import capi
    text = open(os.path.join(os.path.dirname(__file__), "..", "include", "goctr.h")).read()
    m = re.search(r"enum\s*\{\s*(GOCTR_K_ATTN_FWD[^}]*GOCTR_K_COUNT)\s*\}", text, re.S)
    assert m
    names = [x.strip().split("=")[0].strip() for x in m.group(1).split(",") if x.strip()]
    assert names[-1] == "GOCTR_K_COUNT" and capi.PROF_COUNT == len(names) - 1
