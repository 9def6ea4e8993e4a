"""CPU simulation (oracle kernels, no GPU) of the item2vec data-parallel exchange rules: W ranks train their corpus shards from
a common snapshot for K words each, then the parameter deltas are combined.  Which combination keeps the HS loss of a pass at
the sequential / Hogwild level?  (round 3-4's  p = p0 + sum_r (p_r - p0)  once per pass was never measured.)

  sum      p0 + sum_r d_r                      (rounds 3-4)
  mean     p0 + sum_r d_r / W                  (model averaging)
  touched  p0 + sum_r d_r / #{r : d_r[row] != 0}   per ROW: the average over the ranks that updated the row

usage: python scripts/w2v_dp_sim.py [n_words] [W] [K]"""
import sys, os, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import pyoracle as oracle
from test_gpu_fullsize import _session_corpus, _hs_loss

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = int(float(sys.argv[3])) if len(sys.argv) > 3 else 100_000
rng = np.random.default_rng(105)
V, dim = 10681, 16
doc, topics = _session_corpus(rng, V, n)
counts = np.bincount(doc, minlength=V) + 1
p0 = (rng.random((V, dim)) - 0.5) / dim
paths = oracle.huffman_paths(counts)
cfg = oracle.w2v_cfg(dim=dim, optimizer="hs")
sig = oracle.sigmoid_table()
pos = rng.integers(1, n - 1, size=4000)
pairs = list(zip(doc[pos].tolist(), doc[pos + 1].tolist()))
cut = [n * r // W for r in range(W + 1)]
nseg = max(1, -(-(n // W) // K))
print(f"n {n}  W {W}  K {K}  segments {nseg}")

t = time.time()
sp, sa = p0.copy(), np.zeros((V - 1, dim))
oracle.w2v_train_slice(cfg, doc, 0, n, None, sp, sa, paths, sig, oracle.Lcg(1), 0.025, 0, n)
print(f"sequential pass: HS loss {_hs_loss(sp, sa, paths, pairs):.4f}  ({time.time() - t:.0f} s)")

def run(rule, nseg):
    P, A = p0.copy(), np.zeros((V - 1, dim))
    lrs = [0.025] * W
    lcgs = [oracle.Lcg(1 + 7919 * r) for r in range(W)]
    for s in range(nseg):
        def one(r):
            p, a = P.copy(), A.copy()
            lo, hi = cut[r], cut[r + 1]
            a0, a1 = lo + (hi - lo) * s // nseg, lo + (hi - lo) * (s + 1) // nseg
            # the observer counts the other ranks' words too: trained so far ~ W * (a0 - lo)
            lrs[r], _ = oracle.w2v_train_slice(cfg, doc[a0:a1], 0, a1 - a0, None, p, a, paths, sig, lcgs[r], lrs[r], W * (a0 - lo), n)
            return p - P, a - A
        with ThreadPoolExecutor(W) as ex:
            ds = list(ex.map(one, range(W)))
        for M, k in ((P, 0), (A, 1)):
            d = np.stack([x[k] for x in ds])
            if rule == "sum":
                M += d.sum(0)
            elif rule == "mean":
                M += d.sum(0) / W
            else:
                cnt = np.maximum((np.abs(d).max(2) > 0).sum(0), 1)
                M += d.sum(0) / cnt[:, None]
    return _hs_loss(P, A, paths, pairs)

for rule in ("sum", "mean", "touched"):
    for ns in (nseg, 1):
        t = time.time()
        print(f"{rule:8s} segments {ns:3d}: HS loss {run(rule, ns):.4f}  ({time.time() - t:.0f} s)", flush=True)
