"""The claim behind the wave-parallel replay of csrc/search.hip (knn_replay_body): ONE insertion of search.go:104-121 --

    if score > low { carry = (score, item); for i in 0..k-1: if carry.sim > nb[i].sim { swap(carry, nb[i]) }; low = nb[k-1].sim }

-- applied to a non-increasing k-array is exactly: the item lands at p = the first position with score > nb[p]; every later
START of a run of equal similarities takes the FIRST element of the run before it; everything else keeps its place; the first
element of the last run is dropped (the reference's strict `>` for displaced elements: quirk Q22, the tie rotation).  The device
evaluates that rule with two ballots and a shuffle per insertion; here the rule is checked against the literal loop on tie-heavy
sequences (CPU, no device)."""
import numpy as np


def literal(nb_s, nb_i, low, score, it):
    if not score > low:
        return low
    ts, ti = score, it
    for i in range(len(nb_s)):
        if ts > nb_s[i]:
            nb_s[i], ts = ts, nb_s[i]
            nb_i[i], ti = ti, nb_i[i]
    return nb_s[-1]


def rule(nb_s, nb_i, low, score, it):
    k = len(nb_s)
    if not score > low:
        return low
    gt = [score > nb_s[i] for i in range(k)]
    if not any(gt):
        return low
    p = gt.index(True)
    start = [i == p or (i > p and nb_s[i - 1] != nb_s[i]) for i in range(k)]
    old_s, old_i = list(nb_s), list(nb_i)
    for i in range(k):
        if i == p:
            nb_s[i], nb_i[i] = score, it
        elif start[i]:
            src = max(j for j in range(i) if start[j])          # the highest run start below i
            nb_s[i], nb_i[i] = old_s[src], old_i[src]
    return nb_s[-1]


def test_rule_equals_the_literal_insertion_loop():
    rng = np.random.default_rng(0)
    for trial in range(300):
        k = int(rng.integers(1, 12))
        n = int(rng.integers(1, 80))
        pool = rng.random(int(rng.integers(1, 6)))               # few distinct similarities: long tie runs
        sims = np.where(rng.random(n) < 0.7, rng.choice(pool, n), rng.random(n))
        sims[rng.random(n) < 0.1] = 0.0                           # scores that never qualify
        a_s, a_i, la = [0.0] * k, [-1] * k, 0.0
        b_s, b_i, lb = [0.0] * k, [-1] * k, 0.0
        for it in range(n):
            la = literal(a_s, a_i, la, float(sims[it]), it)
            lb = rule(b_s, b_i, lb, float(sims[it]), it)
            assert a_s == b_s and a_i == b_i and la == lb, (trial, it)
