"""Pin the CPU oracle against the reference's own literal known-answer tests
(tests/golden/ref_kats.json, transcribed from model/activation_test.go, model/cost_test.go,
utils/util_test.go, nn/metrics/ranking_test.go) and against constants derivable from the
reference (LCG stream, sigmoid table)."""
import json
import math
import os

import numpy as np
import pytest

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kats.json")))


def test_prelu(oracle):
    for k in KATS["prelu"]:
        out = oracle.prelu32(np.array(k["x"], np.float32), k["slope"])
        # ShouldResemble on []float32: exact
        assert out.tolist() == np.array(k["expect"], np.float32).tolist(), k["src"]


@pytest.mark.parametrize("k", KATS["euc_distance"], ids=lambda k: k["src"])
def test_euc_distance(oracle, k):
    x = np.array(k["x"], np.float32).reshape(k["x_shape"])
    y = np.array(k["y"], np.float32).reshape(k["y_shape"])
    out = oracle.euc_distance(x, y)
    assert list(out.shape) == k["out_shape"]
    assert out.ravel().tolist() == np.array(k["expect"], np.float32).tolist()


@pytest.mark.parametrize("k", KATS["cosine_similarity"], ids=lambda k: k["src"])
def test_cosine_similarity(oracle, k):
    x = np.array(k["x"], np.float32).reshape(k["x_shape"])
    y = np.array(k["y"], np.float32).reshape(k["y_shape"])
    out = oracle.cosine_similarity(x, y)
    assert list(out.shape) == k["out_shape"]
    assert out.ravel().tolist() == np.array(k["expect"], np.float32).tolist()


def test_shape_mismatch_errors(oracle):
    for k in KATS["euc_distance_error"] + KATS["cosine_similarity_error"]:
        x = np.zeros(k["x_shape"], np.float32)
        y = np.zeros(k["y_shape"], np.float32)
        with pytest.raises(ValueError):
            oracle.euc_distance(x, y)
        with pytest.raises(ValueError):
            oracle.cosine_similarity(x, y)


def test_costs(oracle):
    for k in KATS["bce"]:
        assert abs(oracle.bce32(k["y_pred"], k["y_true"]) - k["expect"]) <= k["tol"], k["src"]
    for k in KATS["mse"]:
        assert abs(oracle.mse32(k["y_pred"], k["y_true"]) - k["expect"]) <= k["tol"], k["src"]
    for k in KATS["rms"]:
        assert abs(oracle.rms32(k["y_pred"], k["y_true"]) - k["expect"]) <= k["tol"], k["src"]


def test_bce_epsilon_is_noop(oracle):
    # cost.go:12: float32(1.0+1e-8) == 1.0 exactly (quirk Q2) => p == 1 with y == 0 gives +inf
    assert np.float32(1.0 + 1e-8) == np.float32(1.0)
    assert math.isinf(oracle.bce32([1.0], [0.0]))


def test_roc_auc(oracle):
    for k in KATS["roc_auc"]:
        assert oracle.roc_auc(k["pred"], k["y"]) == k["expect"], k["src"]
        assert oracle.roc_auc32(k["pred"], k["y"]) == pytest.approx(k["expect"], abs=1e-7)
    for k in KATS["roc_curve"]:
        fpr, tpr, thr = oracle.roc_curve(k["scores"], k["y"], k["pos_label"])
        assert fpr.tolist() == k["fpr"] and tpr.tolist() == k["tpr"] and thr.tolist() == k["thresholds"]


def test_roc_auc_matches_sklearn_with_ties(oracle):
    from sklearn.metrics import roc_auc_score
    rng = np.random.default_rng(0)
    for n in (5, 64, 1000):
        s = np.round(rng.random(n), 1)  # many ties
        y = (rng.random(n) < 0.4).astype(np.float64)
        if y.min() == y.max():
            y[0] = 1 - y[0]
        assert oracle.roc_auc(s, y) == pytest.approx(roc_auc_score(y, s), abs=1e-12)


def test_mt19937_matches_numpy():
    # nn/base/source_test.go:18-25: randomkit seed 7 == numpy RandomState(7)
    for k in KATS["mt19937_numpy"]:
        got = np.random.RandomState(k["seed"]).random_sample(len(k["expect"]))
        assert np.allclose(got, k["expect"], atol=5e-9)


def test_lcg_stream(oracle):
    # modelutil.go:21-29: next = next*25214903917 + 11 (mod 2^64), starts at 1
    nxt, exp = 1, []
    for _ in range(64):
        nxt = (nxt * 25214903917 + 11) % (1 << 64)
        exp.append(nxt % 5)
    assert oracle.lcg_stream(64, 5) == exp
    assert exp[:8] == [3, 3, 2, 4, 0, 2, 1, 3]


def test_sigmoid_table(oracle):
    t = oracle.sigmoid_table()
    # sigmoid_table.go:28-45
    for i in (0, 1, 499, 500, 999):
        e = math.exp((i / 1000 * 2. - 1.) * 6.0)
        assert t[i] == e / (e + 1.)
    assert oracle.sigmoid_lookup(t, 3.0) == t[int((3.0 + 6.0) * (1000 / 6.0 / 2.0))]
    assert oracle.sigmoid_lookup(t, 3.0) == pytest.approx(0.9525741268224333, abs=1e-15)
    assert 0 <= oracle.sigmoid_lookup(t, 3.0) <= 1  # sigmoid_table_test.go:21-27


def test_subsample_and_index_per_thread(oracle):
    assert oracle.subsample_keep(1e-3, 1) == 1 - math.sqrt(1e-3)
    assert oracle.subsample_keep(1e-3, 5) == 1 - math.sqrt(1e-3 / 5)
    idx = oracle.index_per_thread(4, 10)
    # modelutil.go:32-41
    exp = [0]
    for i in range(1, 4):
        exp.append(exp[-1] + (10 + i) // 4)
    exp.append(10)
    assert idx.tolist() == exp


# ---------------------------------------------------------------- embedding k-NN search
def test_search_cosine_kat(oracle):
    for c in KATS["search_cosine"]["cases"]:
        assert oracle.cosine64(c["v1"], c["v2"]) == c["expect"]


def _search_case(oracle, case, query, ignore):
    words = [w for w, _ in case["items"]]
    items = np.array([v for _, v in case["items"]], np.float64)
    idx, sim, rank = oracle.knn_search(items, query, case["k"], ignore=ignore)
    got = [{"Word": words[i] if i >= 0 else "", "Rank": int(r), "Similarity": float(s)} for i, s, r in zip(idx, sim, rank)]
    assert got == case["expect"]


def test_search_internal_kat(oracle):
    case = KATS["search_internal"]
    words = [w for w, _ in case["items"]]
    q = words.index(case["word"])
    _search_case(oracle, case, case["items"][q][1], q)


def test_search_vector_kat(oracle):
    case = KATS["search_vector"]
    _search_case(oracle, case, case["query"], -1)


def test_search_tail_quirk_and_ties(oracle):
    # fewer positive-score items than k: the guard loop leaves k-1 entries, the surplus ones empty (search.go:126-131)
    items = np.array([[1, 0], [2, 0], [0, 1], [-1, 0]], np.float64)
    idx, sim, rank = oracle.knn_search(items, [1.0, 0.0], 4)
    assert idx.tolist() == [0, 1, -1] and sim.tolist() == [1.0, 1.0, 0.0] and rank.tolist() == [1, 2, 0]
    # ties keep arrival order (strict > in the bubble-up), non-positive scores never enter
    idx, sim, _ = oracle.knn_search(items, [1.0, 0.0], 2)
    assert idx.tolist() == [0, 1]
    idx, _, _ = oracle.knn_search(items, [1.0, 0.0], 1, ignore=0)
    assert idx.tolist() == [1]
    # a displaced element jumps over its equals (the bubble-up compares with a strict > for it too): inserting a
    # better item above two equal ones rotates them
    idx, _, _ = oracle.knn_search(np.array([[1, 0], [1, 0], [1, 0.1]], np.float64), [1.0, 0.05], 3)
    assert idx.tolist() == [2, 1, 0]


# ---------------------------------------------------------------- user-behaviour cache
def test_ubcache_filter_kats(oracle):
    k = KATS["ubcache_filter"]
    for c in k["cases"]:
        assert oracle.ubcache_filter(k["ts"], k["items"], c["max_ts"], c["max_len"]).tolist() == c["expect"]
    # nothing at or before max_ts -> empty; duplicates of the boundary timestamp are all kept
    assert oracle.ubcache_filter([9, 8, 7], [1, 2, 3], 5, 2).size == 0
    assert oracle.ubcache_filter([9, 7, 7, 7, 3], [1, 2, 3, 4, 5], 7, 2).tolist() == [2, 3]


# ---------------------------------------------------------------- corpus load / dictionary
def test_corpus_dictionary_hand_derived(oracle):
    # dictionary.go:70-81: ids by first appearance, cfs = counts; memory.go:85-88: idoc = id of every word
    keys = [50, 30, 50, 70, 30, 50, -4, 70, 50]
    idoc, id2key, cfs, indexed = oracle.corpus_build(keys, min_count=-1, max_count=-1)
    assert idoc.tolist() == [0, 1, 0, 2, 1, 0, 3, 2, 0]
    assert id2key.tolist() == [50, 30, 70, -4] and cfs.tolist() == [4, 2, 2, 1]
    assert indexed.tolist() == idoc.tolist()                         # both filters off (v < 0 / v <= 0)
    # MinCount(2): drops freq < 2 (cpsutil.go:72-76); MaxCount(3): drops 3 < freq (cpsutil.go:64-68)
    assert oracle.corpus_build(keys, 2, -1)[3].tolist() == [0, 1, 0, 2, 1, 0, 2, 0]
    assert oracle.corpus_build(keys, -1, 3)[3].tolist() == [1, 2, 1, 3, 2]
    assert oracle.corpus_build(keys, 2, 3)[3].tolist() == [1, 2, 1, 2]
    # MinCount(0) is "on" (0 <= v) but drops nothing (freq < 0 never holds); MaxCount(0) is off
    assert oracle.corpus_build(keys, 0, 0)[3].tolist() == idoc.tolist()
    # a filter can empty the doc; the dictionary still holds every word
    i2, k2, c2, x2 = oracle.corpus_build(keys, 100, -1)
    assert x2.size == 0 and k2.size == 4


def test_corpus_matches_python_dict(oracle):
    rng = np.random.default_rng(0)
    keys = rng.zipf(1.3, size=20000) * 7 - 3
    idoc, id2key, cfs, indexed = oracle.corpus_build(keys, 5, 400)
    w2id, counts = {}, []
    ref = []
    for w in keys.tolist():
        if w in w2id:
            counts[w2id[w]] += 1
        else:
            w2id[w] = len(counts)
            counts.append(1)
        ref.append(w2id[w])
    assert idoc.tolist() == ref and cfs.tolist() == counts
    assert [w2id[k] for k in id2key.tolist()] == list(range(len(counts)))
    keep = [i for i in ref if not (counts[i] > 400 or counts[i] < 5)]
    assert indexed.tolist() == keep


def test_subsample_probs(oracle):
    # subsample.go:28-43: z = 1 - sqrt(t / freq), clamped at 0 (raw counts, quirk Q14)
    p = oracle.subsample_probs([1, 4, 1000, 10 ** 6], 1e-3)
    assert p[0] == 1.0 - np.sqrt(1e-3) and p[1] == 1.0 - np.sqrt(1e-3 / 4.0)
    assert p[3] == 1.0 - np.sqrt(1e-3 / 1e6)
    assert oracle.subsample_probs([1, 2], 4.0).tolist() == [0.0, 0.0]


def _nn_forward_theta():
    """packed [b0 | W0 | b1 | W1] (basemlp64.go:432-463) from the KAT's per-neuron input weights (W[in k][out j] =
    weights[layer][neuron j][input k]) for the first two layers of nn/network_test.go's 3-3-3 net"""
    k = KATS["nn_forward"]
    parts = []
    for layer in k["weights_layer_neuron_input"][:2]:
        W = np.asarray(layer, np.float64).T                      # [in, out]
        parts += [np.full(W.shape[1], k["bias"]), W.ravel()]
    return np.concatenate(parts)


def test_nn_forward_kat_pins_the_mlp_forward(oracle):
    """nn/network_test.go:25-83: ReLU layer -> Sigmoid layer of the literal 3-3-3 net = the sklearn-port forward
    (basemlp64.go:259-274) with units [3,3,3], relu hidden, logistic output; the softmax row is checked on top"""
    k = KATS["nn_forward"]
    x = np.asarray([k["input"]], np.float64)
    theta = _nn_forward_theta()
    hid = oracle.mlp_predict(oracle.mlp_cfg([3, 3], "identity"), theta[:12], x)      # logistic(W0 x + 1)
    out = oracle.mlp_predict(oracle.mlp_cfg([3, 3, 3], "relu"), theta, x)[0]
    assert np.allclose(out, k["expected"][1], rtol=k["rel_tol"], atol=0)
    # layer 0 pre-activation through the logistic output: logit(hid) == expected[0] (all positive, so relu is identity)
    assert np.allclose(np.log(hid[0] / (1 - hid[0])), k["expected"][0], rtol=1e-10, atol=0)
    W2 = np.asarray(k["weights_layer_neuron_input"][2], np.float64).T
    z = out @ W2 + k["bias"]
    sm = np.exp(z - z.max()); sm /= sm.sum()
    assert np.allclose(sm, k["expected"][2], rtol=k["rel_tol"], atol=0)
