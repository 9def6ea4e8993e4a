"""CPU check of the error bound the k-NN scan path rests on (csrc/search.hip): the float32 filter's score
a = sum_d fl32(q_d / |q|) * fl32(v_d / |v|), accumulated in float32 in ANY order, differs from the reference's float64 cosine
(searchutil.go:17-26) by at most E = (D + 8) * 2^-23 -- whatever the vectors' magnitudes, because both operands are normalised
before they are rounded.  The device kernels (packed FMAs or the float32 MFMA) only change the accumulation order."""
import numpy as np
import pytest


def cosine64(q, v):
    dot = 0.0
    for a, b in zip(q, v):
        dot += a * b                      # the reference's d-order
    qn, vn = np.sqrt((q * q).sum()), np.sqrt((v * v).sum())
    return dot / qn / vn


@pytest.mark.parametrize("D", [16, 32, 64])
def test_float32_filter_error_is_inside_E(D):
    rng = np.random.default_rng(D)
    E = (D + 8) * 2.0 ** -23
    worst = 0.0
    for trial in range(400):
        scale_q, scale_v = 10.0 ** rng.uniform(-150, 150, size=2)          # magnitudes far outside float32's range
        q = rng.standard_normal(D) * scale_q
        if trial % 3 == 0:                                                 # near-parallel pairs: similarities close to 1
            v = (q / scale_q + 1e-3 * rng.standard_normal(D)) * scale_v
        elif trial % 3 == 1:                                               # a few large components among tiny ones
            v = rng.standard_normal(D) * scale_v * np.where(rng.random(D) < 0.2, 1.0, 1e-9)
        else:
            v = rng.standard_normal(D) * scale_v
        sim = cosine64(q, v)
        q32 = (q / np.sqrt((q * q).sum())).astype(np.float32)
        v32 = (v / np.sqrt((v * v).sum())).astype(np.float32)
        prods = q32 * v32                                                  # float32 products
        for order in (np.arange(D), np.arange(D)[::-1], rng.permutation(D)):
            acc = np.float32(0.0)
            for d in order:
                acc = np.float32(acc + prods[d])
            worst = max(worst, abs(float(acc) - sim))
            # two accumulators over even / odd dimensions (the packed-FMA layout of knn_scan_kernel)
        acc2 = np.float32(prods[0::2].sum(dtype=np.float32) + prods[1::2].sum(dtype=np.float32))
        worst = max(worst, abs(float(acc2) - sim))
    assert worst <= E, (worst, E)
    assert worst > 0                                                        # (the filter really is approximate)


def test_candidate_set_argument():
    """The replay argument on a small instance: inserting ANY superset of C = {sim >= k-th best} in item order, with the reference's
    strict-> insertion (search.go:104-121), leaves the k-array the full loop leaves -- ties included."""
    rng = np.random.default_rng(3)

    def insert_all(sims, order, k):
        nb_s, nb_i = [0.0] * k, [-1] * k
        for it in order:
            s = sims[it]
            if not s > nb_s[k - 1]:
                continue
            ts, ti = s, it
            for i in range(k):
                if ts > nb_s[i]:
                    nb_s[i], ts = ts, nb_s[i]
                    nb_i[i], ti = ti, nb_i[i]
        return nb_s, nb_i

    for trial in range(200):
        n, k = int(rng.integers(5, 60)), int(rng.integers(1, 8))
        sims = np.round(rng.random(n), 1 if trial % 2 else 3)               # coarse values: many exact ties
        sims[rng.random(n) < 0.2] = 0.0
        full = insert_all(sims, range(n), k)
        pos = np.sort(sims[sims > 0])[::-1]
        thr = pos[k - 1] if pos.size >= k else 0.0
        C = [i for i in range(n) if sims[i] > 0 and sims[i] >= thr]
        extra = [i for i in range(n) if i not in C and rng.random() < 0.5]  # a random superset
        S = sorted(C + extra)
        assert insert_all(sims, S, k) == full
        assert insert_all(sims, sorted(C), k) == full


def _bf16(x):
    """float32 array -> the nearest bfloat16 (ties to even), back as float32: the device's (__bf16) cast"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return (u & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("D", [16, 32])
def test_bf16_plane_filter_error_is_inside_2_pow_minus_14(D):
    """knn_scan_bf16_kernel (round 5): rows and queries as two bf16 planes, x = hi + lo + rest, score = hi.hi + hi.lo + lo.hi with
    float32 accumulation (v_mfma_f32_32x32x16_bf16: the products of bf16 pairs are exact in float32).  |score - cosine| <= 2^-14 for
    any accumulation order of the 3 D products, whatever the vectors' magnitudes."""
    rng = np.random.default_rng(100 + D)
    E = 2.0 ** -14
    worst = 0.0
    for trial in range(400):
        scale_q, scale_v = 10.0 ** rng.uniform(-150, 150, size=2)
        q = rng.standard_normal(D) * scale_q
        if trial % 3 == 0:
            v = (q / scale_q + 1e-3 * rng.standard_normal(D)) * scale_v
        elif trial % 3 == 1:
            v = rng.standard_normal(D) * scale_v * np.where(rng.random(D) < 0.2, 1.0, 1e-9)
        else:
            v = rng.standard_normal(D) * scale_v
        sim = cosine64(q, v)
        q32 = (q / np.sqrt((q * q).sum())).astype(np.float32)
        v32 = (v / np.sqrt((v * v).sum())).astype(np.float32)
        qh = _bf16(q32); ql = _bf16(q32 - qh)
        vh = _bf16(v32); vl = _bf16(v32 - vh)
        assert np.all(np.abs(q32 - qh - ql) <= 2.0 ** -16 * np.abs(q32) + 1e-45)      # the two-plane split's residual
        terms = np.concatenate([vl * qh, vh * ql, vh * qh]).astype(np.float32)        # (each product exact in float32)
        for order in (np.arange(3 * D), np.arange(3 * D)[::-1], rng.permutation(3 * D)):
            acc = np.float32(0.0)
            for i in order:
                acc = np.float32(acc + terms[i])
            worst = max(worst, abs(float(acc) - sim))
    assert 0 < worst <= E, (worst, E)
