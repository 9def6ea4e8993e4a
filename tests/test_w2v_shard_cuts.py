"""goctr_w2v_shard_cuts (host-only entry of the device library: callable without a GPU): the word ranges a single-call
multi-device item2vec pass (goctr_w2v_cfg.devices) gives its ranks are whole groups of the reference's goroutine slices
(IndexPerThread, modelutil.go:32-41) -- so every window is clipped where the reference clips it."""
import ctypes as C

import numpy as np
import pytest


def cuts(n, slices, devices):
    from goctr_amd import capi
    out = np.zeros(devices + 1, np.int64)
    capi.check(capi.load().goctr_w2v_shard_cuts(C.c_int64(n), C.c_int(slices), C.c_int(devices), capi.ptr(out, C.c_int64)))
    return out


@pytest.mark.parametrize("n", [16, 1000, 10_000_001, 123_456_789])
@pytest.mark.parametrize("devices", [1, 2, 4, 8])
def test_cuts_are_slice_boundaries(n, devices):
    from oracle import pyoracle
    S = 16
    idx = pyoracle.index_per_thread(S, n)                       # the oracle's IndexPerThread
    c = cuts(n, S, devices)
    assert c[0] == 0 and c[-1] == n and np.all(np.diff(c) > 0)
    assert np.array_equal(c, idx[:: S // devices])              # rank r = slices [r, r + 1) * S / devices


def test_uneven_slices_fall_back_to_equal_ranges():
    c = cuts(1000, 6, 4)                                        # 6 slices do not divide over 4 ranks
    assert list(c) == [0, 250, 500, 750, 1000]
    assert list(cuts(1000, 0, 2)) == [0, 500, 1000]             # slices = 0: one slice per worker
    from goctr_amd import capi
    out = np.zeros(9, np.int64)
    assert capi.load().goctr_w2v_shard_cuts(C.c_int64(3), C.c_int(16), C.c_int(8), capi.ptr(out, C.c_int64)) != 0    # fewer words than ranks
