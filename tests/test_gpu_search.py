"""GPU parity for the embedding k-NN searcher: indices, similarities (float64) and the returned
slice length must equal the oracle's sequential restatement of search.go:92-134 BIT FOR BIT, through the C ABI."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kats.json")))


def test_reference_kats():
    from goctr_amd import search as gs
    case = KATS["search_internal"]
    s = gs.New(*[(w, v) for w, v in case["items"]])
    got = s.SearchInternal(case["word"], case["k"])
    assert [vars(n) for n in got] == case["expect"]
    case = KATS["search_vector"]
    got = gs.New(*[(w, v) for w, v in case["items"]]).SearchVector(case["query"], case["k"])
    assert [vars(n) for n in got] == case["expect"]
    with pytest.raises(KeyError):
        s.SearchInternal("zebra", 1)


@pytest.mark.parametrize("V,D,k", [(1, 4, 3), (37, 5, 1), (300, 16, 10), (5000, 16, 25), (4097, 64, 256), (20000, 10, 7)])
def test_matches_oracle_bit_exact(oracle, V, D, k):
    from goctr_amd import search as gs
    rng = np.random.default_rng(V + D + k)
    items = rng.standard_normal((V, D))
    items[rng.random(V) < 0.05] = 0.0                       # zero-norm items score 0 and never qualify
    dup = rng.integers(0, V, size=max(V // 10, 1))          # exact duplicates: ties resolved by arrival order
    items[dup] = items[rng.integers(0, V, size=dup.size)]
    s = gs.Searcher([f"w{i}" for i in range(V)], items)
    Q = 9
    queries = rng.standard_normal((Q, D))
    queries[0] = items[min(3, V - 1)]                       # a query that is one of the items
    queries[1] = 0.0                                        # zero query: nothing qualifies
    ignore = np.full(Q, -1, np.int64)
    ignore[0] = min(3, V - 1)
    ignore[2] = 0
    idx, sim, cnt = s.search_vectors(queries, k, ignore)
    for q in range(Q):
        ri, rs, _ = oracle.knn_search(items, queries[q], k, ignore=int(ignore[q]))
        assert cnt[q] == ri.size
        assert np.array_equal(idx[q, :cnt[q]], ri)
        assert np.array_equal(sim[q, :cnt[q]], rs)          # bit-exact float64


@pytest.mark.parametrize("V,D,k,Q", [(30000, 16, 10, 70), (2500, 32, 40, 33), (1024, 8, 5, 1)])
def test_scan_path_query_blocks(oracle, V, D, k, Q):
    """the scan path (csrc/search.hip: tile maxima of a division-free score -> bound -> exact similarities of the few
    candidates -> replay) over several 32-query blocks and tiles, with near-duplicate items whose similarities differ in the
    last bits (the bound's 1e-12 margin must keep every member of the candidate set) and ignored items that own a tile's
    maximum; the same call with GOCTR_KNN_SCAN=0 (the tile kernels) must give the same bits"""
    from goctr_amd import search as gs
    rng = np.random.default_rng(V + D + k + Q)
    items = rng.standard_normal((V, D))
    near = rng.integers(0, V, size=V // 20)
    items[near] = items[rng.integers(0, V, size=near.size)] * (1.0 + rng.integers(-3, 4, size=(near.size, 1)) * 2.0 ** -52)
    s = gs.Searcher([str(i) for i in range(V)], items)
    queries = rng.standard_normal((Q, D))
    own = rng.integers(0, V, size=Q)
    queries[::3] = items[own[::3]]                          # a third of the queries are items: their own row is ignored
    ignore = np.full(Q, -1, np.int64)
    ignore[::3] = own[::3]
    idx, sim, cnt = s.search_vectors(queries, k, ignore)
    for q in range(Q):
        ri, rs, _ = oracle.knn_search(items, queries[q], k, ignore=int(ignore[q]))
        assert cnt[q] == ri.size
        assert np.array_equal(idx[q, :cnt[q]], ri) and np.array_equal(sim[q, :cnt[q]], rs)
    # the tile kernels; the VALU / matrix-core scan kernels forced;
    # the host waiting on the stream for every call / watching the pinned counts for every call (default: up to 64 queries);
    # the input through a staged copy instead of host stores into device memory over the BAR
    for var, val in (("GOCTR_KNN_SCAN", "0"), ("GOCTR_KNN_MFMA", "1"), ("GOCTR_KNN_MFMA", "0"),
                     ("GOCTR_KNN_POLL_MAXQ", "0"), ("GOCTR_KNN_POLL_MAXQ", "100000"), ("GOCTR_KNN_BAR", "0")):
        os.environ[var] = val
        try:
            idx0, sim0, cnt0 = s.search_vectors(queries, k, ignore)
        finally:
            del os.environ[var]
        assert np.array_equal(idx0, idx) and np.array_equal(sim0, sim) and np.array_equal(cnt0, cnt), (var, val)


@pytest.mark.parametrize("V,k,D", [(5000, 3, 8), (5000, 25, 8), (9000, 256, 8), (9000, 25, 16), (40000, 10, 16)])
def test_heavy_ties_replay(oracle, V, k, D):
    """items drawn from a pool of a few distinct vectors: thousands of exactly equal similarities, tie groups cut by
    the k-th place and by the per-tile lists -- the order inside a tie group depends on the arrival history
    (search.go:108-115 uses a strict > for displaced elements too) and must still match bit for bit"""
    from goctr_amd import search as gs
    rng = np.random.default_rng(V + k)
    pool = rng.standard_normal((6, D))
    items = pool[rng.integers(0, 6, size=V)]
    items[rng.random(V) < 0.02] = rng.standard_normal(D)            # a few singletons in between
    s = gs.Searcher([str(i) for i in range(V)], items)              # (D = 16: the scan path's candidate lists overflow -> tile kernels)
    queries = rng.standard_normal((5, D))
    queries[0] = pool[0]
    idx, sim, cnt = s.search_vectors(queries, k)
    for q in range(5):
        ri, rs, _ = oracle.knn_search(items, queries[q], k)
        assert cnt[q] == ri.size
        assert np.array_equal(idx[q, :cnt[q]], ri) and np.array_equal(sim[q, :cnt[q]], rs)


def test_scan_path_few_positive_items(oracle):
    """fewer than k items with a positive similarity (and fewer than k positive group maxima): the bound is 0, every positive
    item is a candidate, the returned slice is the reference's short one"""
    from goctr_amd import search as gs
    rng = np.random.default_rng(77)
    V, D, k = 3000, 16, 10
    base = rng.standard_normal(D)
    items = -np.abs(rng.standard_normal((V, 1))) * base + 1e-3 * rng.standard_normal((V, D))   # almost all opposite to `base`
    pos = rng.choice(V, size=6, replace=False)
    items[pos] = base * rng.random((6, 1)) + 1e-3 * rng.standard_normal((6, D))
    s = gs.Searcher([str(i) for i in range(V)], items)
    queries = np.stack([base, -base, base * 1e-150, np.zeros(D)])
    idx, sim, cnt = s.search_vectors(queries, k)
    for q in range(4):
        ri, rs, _ = oracle.knn_search(items, queries[q], k)
        assert cnt[q] == ri.size
        assert np.array_equal(idx[q, :cnt[q]], ri) and np.array_equal(sim[q, :cnt[q]], rs)


def test_tail_quirk_and_mirror_objects(oracle):
    from goctr_amd import search as gs
    s = gs.Searcher(["a", "b", "c", "d"], [[1, 0], [2, 0], [0, 1], [-1, 0]])
    nb = s.SearchVector([1.0, 0.0], 4)                      # only 2 items qualify: 3 entries come back, the last empty
    assert [(n.Word, n.Rank, n.Similarity) for n in nb] == [("a", 1, 1.0), ("b", 2, 1.0), ("", 0, 0.0)]
    assert [n.Word for n in s.SearchInternal("a", 1)] == ["b"]


def test_large_scan_properties():
    """full-size property check (V = 10^6): every returned similarity is >= every non-returned one"""
    from goctr_amd import search as gs
    rng = np.random.default_rng(3)
    V, D, k = 1_000_000, 16, 20
    items = rng.standard_normal((V, D))
    s = gs.Searcher([str(i) for i in range(V)], items)
    q = rng.standard_normal((2, D))
    idx, sim, cnt = s.search_vectors(q, k)
    for j in range(2):
        scores = items @ q[j] / np.linalg.norm(q[j]) / np.linalg.norm(items, axis=1)
        top = np.sort(scores)[::-1][:k]
        assert cnt[j] == k and np.all(np.diff(sim[j]) <= 0)
        assert np.allclose(sim[j], top, rtol=1e-12, atol=0)
        assert np.allclose(scores[idx[j]], sim[j], rtol=1e-12)


def test_bench_shape_matches_oracle_bit_exact(oracle):
    """the shape bench.py --workload knn times (V = 10^6, D = 16, k = 10, the same seeded items): 64 queries per call ->
    knn_scan_bf16_kernel<16> over 977 tiles -> knn_collect (+ replay).  Q = 11 is the last call size on the VALU scan kernel, 12 the
    first on the matrix-core one, 64 a full query block (and the last size whose completion the host polls), 65 two blocks; each
    call also forced onto the other scan kernel.  Indices, float64 similarities and counts equal the oracle's sequential loop for
    EVERY query."""
    from goctr_amd import search as gs
    rng = np.random.default_rng(42)
    V, D, k = 1_000_000, 16, 10
    items = rng.standard_normal((V, D))
    s = gs.Searcher([""] * V, items)
    acc = np.zeros(V)
    for d in range(D):                                      # embutil.go:21-27 / orc_norm64: the d-ordered sum, then sqrt
        acc += items[:, d] * items[:, d]
    norms = np.sqrt(acc)
    assert all(norms[i] == oracle.norm64(items[i]) for i in range(0, V, 9973))
    allq = rng.standard_normal((65, D))
    allq[5] = items[123456]                                 # a query that is an item; its own row ignored
    ign = np.full(65, -1, np.int64)
    ign[5] = 123456
    want = [oracle.knn_search(items, allq[q], k, ignore=int(ign[q]), norms=norms) for q in range(65)]
    for Q in (11, 12, 47, 64, 65):
        # default dispatch (>= 12 queries: the bf16-plane matrix-core filter), the VALU filter, the matrix-core filter
        for mfma in (None, "0", "1"):
            bf16 = None
            if mfma is not None:
                os.environ["GOCTR_KNN_MFMA"] = mfma
            try:
                idx, sim, cnt = s.search_vectors(allq[:Q], k, ign[:Q])
            finally:
                os.environ.pop("GOCTR_KNN_MFMA", None)
            for q in range(Q):
                ri, rs, _ = want[q]
                assert cnt[q] == ri.size == k, (Q, mfma, bf16, q)
                assert np.array_equal(idx[q, :k], ri) and np.array_equal(sim[q, :k], rs), (Q, mfma, bf16, q)
