"""The Huffman build WITH the device (csrc/huffman.hip: rocPRIM stable sort by count, the two-queue merge on the host in
sorted-rank space, chain lengths + prefix sum + root-first path fill on the device) must give, bit for bit, the paths of the
host-only builder (csrc/w2v.hip build_huffman), which tests/test_huffman_scale.py ties to the literal O(V^2) restatement of
dictionary/huffman.go:23-57 and node/node.go:39-42."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def both(counts, max_depth=100):
    from goctr_amd import capi, embedding as ge
    capi.init(0)
    res = []
    for dev in ("0", "1"):
        os.environ["GOCTR_HUFFMAN_DEVICE"] = dev
        try:
            res.append(ge.huffman_paths(counts, max_depth=max_depth))
        finally:
            os.environ.pop("GOCTR_HUFFMAN_DEVICE", None)
    return res


@pytest.mark.parametrize("V,hi", [(1, 5), (2, 2), (3, 2), (7, 3), (1000, 3), (100_000, 40), (100_000, 2), (300_001, 1000)])
def test_device_builder_equals_host_builder_on_tie_heavy_counts(V, hi):
    rng = np.random.default_rng(V + hi)
    counts = rng.integers(1, hi, size=V).astype(np.int64)
    host, dev = both(counts)
    for a, b in zip(host, dev):
        assert np.array_equal(a, b)


def test_device_builder_zipf_at_cfg5_stress_size_and_clamp():
    V = 1_000_000
    counts = np.maximum(1, (2e7 / np.arange(1, V + 1)).astype(np.int64))
    np.random.default_rng(7).shuffle(counts)
    host, dev = both(counts)
    for a, b in zip(host, dev):
        assert np.array_equal(a, b)
    depth = np.diff(dev[0])
    assert abs(np.sum(np.exp2(-depth.astype(np.float64))) - 1.0) < 1e-9      # Kraft equality: a full binary tree, nothing clamped
    lin = (2 ** np.arange(40)).astype(np.int64)                               # a degenerate (linear) tree: depths up to 39, clamped at 10
    host, dev = both(lin, max_depth=10)
    for a, b in zip(host, dev):
        assert np.array_equal(a, b)


def test_item2vec_model_built_on_the_device_trains_like_the_host_built_one():
    """goctr_w2v_create above the device threshold: the paths are born in HBM (fetched lazily by get_paths) and a
    deterministic training pass gives the same bits as with host-built paths"""
    from goctr_amd import capi, embedding as ge
    capi.init(0)
    rng = np.random.default_rng(3)
    V, n, dim = 3000, 20000, 16
    p = 1.0 / np.arange(1, V + 1); p /= p.sum()
    doc = rng.choice(V, size=n, p=p).astype(np.int32)
    counts = np.bincount(doc, minlength=V) + 1
    p0 = (rng.random((V, dim)) - 0.5) / dim
    out = []
    for dev in ("0", "1"):
        os.environ["GOCTR_HUFFMAN_DEVICE"] = dev
        try:
            m = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=True)
            m.create(counts, p0)
        finally:
            os.environ.pop("GOCTR_HUFFMAN_DEVICE", None)
        m.train_pass(doc, n, None, lr=0.025)
        out.append((m.get_param(), m.get_aux(), m.get_paths()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    for a, b in zip(out[0][2], out[1][2]):
        assert np.array_equal(a, b)
