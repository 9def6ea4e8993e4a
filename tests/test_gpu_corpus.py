"""GPU parity for the device-side corpus load: dictionary ids / counts / indexed doc must equal
the oracle's restatement of memory.go:53-102 + dictionary.go:70-81 + cpsutil.go:58-78 BIT FOR BIT (index work), at
every size, whatever order the device's atomics land in."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(oracle, keys, min_count, max_count, batches=1):
    from goctr_amd.corpus import Corpus
    keys = np.asarray(keys, np.int64)
    c = Corpus(keys.size + 3, min_count, max_count)
    c.Load(np.array_split(keys, batches))
    idoc, id2key, cfs, indexed = oracle.corpus_build(keys, min_count, max_count)
    assert (c.Len(), c.V, c.n_indexed) == (keys.size, id2key.size, indexed.size)
    gk, gc = c.Dictionary()
    assert np.array_equal(gk, id2key) and np.array_equal(gc, cfs)
    assert np.array_equal(c.idoc(), idoc)
    assert np.array_equal(c.IndexedDoc(), indexed)
    return c


def test_hand_derived_kat(oracle):
    c = _check(oracle, [50, 30, 50, 70, 30, 50, -4, 70, 50], 2, 3)
    assert c.IndexedDoc().tolist() == [1, 2, 1, 2] and c.Dictionary()[1].tolist() == [4, 2, 2, 1]


@pytest.mark.parametrize("n,vocab,mn,mx,batches", [
    (1, 1, -1, -1, 1), (2, 1, 5, -1, 1), (257, 3, 0, 0, 2), (4096, 4096, -1, -1, 1), (4097, 50, 5, 200, 3),
    (100_000, 3000, 5, -1, 7), (1_000_003, 200_000, 5, 5000, 4)])
def test_matches_oracle(oracle, n, vocab, mn, mx, batches):
    rng = np.random.default_rng(n)
    p = 1.0 / np.arange(1, vocab + 1) ** 1.1
    keys = rng.choice(vocab, size=n, p=p / p.sum()).astype(np.int64)
    keys = keys * 1_000_003 - 17 * (keys % 5)                            # spread over the int64 range, some negative
    _check(oracle, keys, mn, mx, batches)


def test_all_distinct_and_all_equal(oracle):
    _check(oracle, np.arange(70_000)[::-1] * 2**33, -1, -1)              # V = n, ids = positions
    _check(oracle, np.full(70_000, 2**62), 5, -1)                        # V = 1, one hot slot


def test_errors():
    from goctr_amd import capi
    from goctr_amd.corpus import Corpus
    c = Corpus(4)
    with pytest.raises(capi.GoctrError):
        c.append([1, 2, 3, 4, 5])                                        # over capacity
    with pytest.raises(capi.GoctrError):
        c.append([np.iinfo(np.int64).min])                               # reserved token
    with pytest.raises(capi.GoctrError):
        c.build()                                                        # empty corpus


def test_subsample_mask_and_resident_training(oracle):
    """keep mask = samples[id] > u with samples bit-equal to subsample.go:28-43; the resident pipeline trains the same
    bits as handing the same doc + mask to the (already pinned) goctr_w2v_train entry."""
    from goctr_amd import embedding as ge
    rng = np.random.default_rng(5)
    n, vocab, dim = 60_000, 400, 16
    p = 1.0 / np.arange(1, vocab + 1) ** 1.2
    keys = rng.choice(vocab, size=n, p=p / p.sum()).astype(np.int64) * 13 + 1000
    thr = 50.0                                                           # raw-count subsampling bites from freq > 50
    m = ge.Word2Vec(dim=dim, iter=1, min_count=5, subsample_threshold=thr, deterministic=True,
                    rng=np.random.default_rng(9))
    m.TrainIds(np.array_split(keys, 3), n, seed=77)
    idoc, id2key, cfs, indexed = oracle.corpus_build(keys, 5, -1)
    assert m.V == cfs.size
    keep = m.keep_mask(indexed.size)
    probs = oracle.subsample_probs(cfs, thr)
    # the mask is consistent with the table: never kept where samples == 0, rate tracks samples elsewhere
    assert not keep[probs[indexed] == 0.0].any()
    hot = probs[indexed] > 0.5
    assert hot.sum() > 1000 and abs(keep[hot].mean() - probs[indexed][hot].mean()) < 0.02
    # the same mask again from the same seed (reproducible), a different one from another seed
    m2 = ge.Word2Vec(dim=dim, iter=1, min_count=5, subsample_threshold=thr, deterministic=True,
                     rng=np.random.default_rng(9))
    m2.TrainIds([keys], n, seed=77)
    assert np.array_equal(m2.keep_mask(indexed.size), keep)
    assert np.array_equal(m2.get_param(), m.get_param())
    # explicit route with the same init, doc and mask
    m3 = ge.Word2Vec(dim=dim, deterministic=True)
    p0 = (np.random.default_rng(9).random((m.V, dim)) - 0.5) / dim
    m3.create(cfs, p0)
    m3.train_pass(indexed, n, keep, lr=0.025)
    assert np.array_equal(m3.get_param(), m.get_param()) and np.array_equal(m3.get_aux(), m.get_aux())
