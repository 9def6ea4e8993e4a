"""Huffman tree / paths of the item2vec path at scale (dictionary/huffman.go:23-57, node/node.go:39-42).

The product builds the tree on the host (goctr_huffman_build: two queues, O(V log V)); round 2 checked it against the
literal O(V^2) restatement of the reference's insertion loop only up to V = 999, while BASELINE configs[4]'s stress point is
V = 10^6.  Host code on both sides: no GPU needed."""
import numpy as np
import pytest

from goctr_amd import embedding as ge


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("V,hi", [(1, 5), (2, 2), (3, 2), (1000, 3), (100_000, 40), (100_000, 2)])
def test_product_builder_equals_the_literal_restatement_on_tie_heavy_counts(oracle, V, hi):
    """counts drawn from [1, hi): thousands of equal counts, so every tie-breaking rule of huffman.go is exercised (leaves
    stable-sorted by count; a merged node goes in FRONT of every queued node of equal value)"""
    rng = np.random.default_rng(V + hi)
    counts = rng.integers(1, hi, size=V).astype(np.int64)
    got = ge.huffman_paths(counts)
    assert _same(got, oracle.huffman_paths(counts, slow=True))


def test_zipf_counts_at_cfg5_stress_size(oracle):
    """V = 10^6 Zipf(1.0) counts (the long tail is one huge tie at count 1): product == the oracle's fast builder, which the
    test above ties to the literal restatement; depth clamp max_depth = 100 like GetPath"""
    V = 1_000_000
    rng = np.random.default_rng(7)
    counts = np.maximum(1, (2e7 / np.arange(1, V + 1)).astype(np.int64))
    rng.shuffle(counts)
    got = ge.huffman_paths(counts)
    ref = oracle.huffman_paths(counts)
    assert _same(got, ref)
    off = got[0]
    depth = np.diff(off)
    assert depth.max() <= 99 and depth.min() >= 1
    # Kraft equality of a full binary tree (no path was clamped here): sum 2^-depth == 1
    assert abs(np.sum(np.exp2(-depth.astype(np.float64))) - 1.0) < 1e-9


def test_max_depth_clamp_matches(oracle):
    counts = (2 ** np.arange(40)).astype(np.int64)            # a degenerate (linear) tree: depths up to 39
    assert _same(ge.huffman_paths(counts, max_depth=10), oracle.huffman_paths(counts, max_depth=10, slow=True))
