"""CPU tests of round-5 host logic (no device): the MLP wrapper hands EVERY row to goctr_mlp_fit (rounds 1-4 cut the rows to a
multiple of the batch: quirk Q11 was sidestepped, now it is reproduced on the device), and the segment arithmetic of the
data-parallel item2vec pass (csrc/w2v.hip: launch `seg` of `nseg` walks positions [len seg / nseg, len (seg + 1) / nseg) of every
stream's piece) partitions every piece exactly."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r'''
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
const char* goctr_last_error(void) { return "stub"; }
size_t goctr_mlp_nparams(const void* p) { (void)p; return 4; }
int goctr_mlp_fit(void* p, const float* X, const float* Y, int64_t rows, const int32_t* perm, double* curve, int* iters) {
  (void)p; (void)X; (void)Y; (void)curve;
  FILE* f = fopen(getenv("STUB_OUT"), "w");
  /* the last permutation entry of the first epoch must be addressable: perm is [max_iter][rows] */
  fprintf(f, "%lld %d\n", (long long)rows, perm ? perm[rows - 1] : -1);
  fclose(f);
  *iters = 1;
  return 0;
}
'''


def test_mlp_fit_passes_every_row(tmp_path):
    sys.path.insert(0, ROOT)
    from goctr_amd import capi
    src = tmp_path / "stub.c"
    names = {"goctr_last_error", "goctr_mlp_nparams", "goctr_mlp_fit"}
    src.write_text(STUB + "".join(f"int {s}() {{ return 0; }}\n" for s in capi.SYMBOLS if s not in names))
    so = tmp_path / "libstub.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-w", str(src), "-o", str(so)], check=True)
    out = tmp_path / "rows.txt"
    code = f'''
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
from goctr_amd import mlp as gmlp
rng = np.random.default_rng(0)
X = rng.random((79948 // 100, 7)).astype(np.float32)           # 799 rows at batch 200: three whole batches + 199
Y = (rng.random(X.shape[0]) < 0.5).astype(np.float32)
clf = gmlp.MLPClassifier([5], "relu", "adam", 1e-5)
clf.BatchSize, clf.MaxIter, clf.RandomState = 200, 2, np.random.default_rng(1)
clf.Fit(X, Y)
'''
    env = dict(os.environ, GOCTR_LIB=str(so), STUB_OUT=str(out))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    rows, last = out.read_text().split()
    assert int(rows) == 799                                        # not 600
    assert 0 <= int(last) < 799


@pytest.mark.parametrize("n_words,streams,nseg", [(1_250_000, 610, 13), (1000, 7, 13), (97, 97, 4), (50_000, 3, 1)])
def test_w2v_segments_partition_every_piece(n_words, streams, nseg):
    # the pieces as run_pass cuts them (one slice: idx[g] = n_words g / streams), then the kernel's per-launch part of a piece
    idx = [n_words * g // streams for g in range(streams + 1)]
    for g in range(streams):
        lo0, len0 = idx[g], idx[g + 1] - idx[g]
        covered = []
        for seg in range(nseg):
            pb, pe = len0 * seg // nseg, len0 * (seg + 1) // nseg
            covered += list(range(lo0 + pb, lo0 + pe))
        assert covered == list(range(idx[g], idx[g + 1]))
    # the host's per-launch loop bound is the longest part of any piece
    for seg in range(nseg):
        mx = max((idx[g + 1] - idx[g]) * (seg + 1) // nseg - (idx[g + 1] - idx[g]) * seg // nseg for g in range(streams))
        assert mx <= -(-max(idx[g + 1] - idx[g] for g in range(streams)) // nseg) + 1
