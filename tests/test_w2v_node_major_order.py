"""CPU check of the claim behind the node-major Hogwild kernel (csrc/w2v.hip: w2v_hogwild_nm_kernel): walking the pairs
of a position JB at a time, node by node -- a node vector read once per chunk, pair j + 1 reading pair j's update from a local
copy, the chunk's summed update written once, a context word that repeats inside a window starting a new chunk -- is the
SEQUENTIAL pair-major arithmetic of the reference (model.go:48-78, optimizer.go:107-129) up to float64 rounding of the summed
update.  Pair-major side: the oracle's C restatement (bit-exact vs the reference's algorithm); node-major side: numpy, here."""
import numpy as np
import pytest


def node_major_pass(pyo, doc, keep, param, aux, paths, tab, cfg_dim, win, lr, init_lr, min_lr, ulb, corpus_len, JB):
    off, nodes, codes = paths
    n = doc.size
    lcg = pyo.Lcg(1)
    L = pyo.lib()
    import ctypes as C
    cnt = 0
    for pos in range(n):
        if keep is None or keep[pos]:
            wid = int(doc[pos])
            dl = L.orc_lcg_next(C.byref(lcg), C.c_int(win))
            ctxs = [int(doc[pos - win + a]) for a in range(dl, 2 * win + 1 - dl) if a != win and 0 <= pos - win + a < n]
            path = [(int(nodes[i]), int(codes[i])) for i in range(off[wid], off[wid + 1])]
            i = 0
            while i < len(ctxs):
                chunk = []
                while i < len(ctxs) and len(chunk) < JB and ctxs[i] not in chunk:
                    chunk.append(ctxs[i]); i += 1
                ctx = [param[c].copy() for c in chunk]
                tmp = [np.zeros(cfg_dim) for _ in chunk]
                alive = [True] * len(chunk)
                for nd, code in path:
                    if not any(alive):
                        break
                    pv = aux[nd].copy()
                    acc = np.zeros(cfg_dim)
                    for j in range(len(chunk)):
                        if not alive[j]:
                            continue
                        inner = 0.0
                        for d in range(cfg_dim):
                            inner += ctx[j][d] * pv[d]
                        if inner <= -6.0 or inner >= 6.0:
                            alive[j] = False                      # quirk Q13: this pair's walk ends
                            continue
                        g = (1.0 - code - tab[int((inner + 6.0) * (1000.0 / 6.0 / 2.0))]) * lr
                        tmp[j] += g * pv
                        pv += g * ctx[j]
                        acc += g * ctx[j]
                    aux[nd] += acc
                for j, c in enumerate(chunk):
                    param[c] += tmp[j]
        cnt += 1
        if cnt % ulb == 0:
            lr = min_lr if lr < min_lr else init_lr * (1.0 - cnt / corpus_len)
    return lr


@pytest.mark.parametrize("JB", [1, 2, 4, 10])
def test_node_major_chunks_equal_the_pair_major_pass(JB):
    from oracle import pyoracle as pyo
    rng = np.random.default_rng(JB)
    V, n, dim, win = 40, 700, 8, 5
    p = 1.0 / np.arange(1, V + 1) ** 1.2
    p /= p.sum()
    doc = rng.choice(V, size=n, p=p).astype(np.int32)
    doc[50:54] = 3                                                  # the same word on both sides of a centre
    counts = np.bincount(doc, minlength=V) + 1
    keep = (rng.random(n) < 0.85).astype(np.uint8)
    paths = pyo.huffman_paths(counts)
    tab = pyo.sigmoid_table()
    p0 = (rng.random((V, dim)) - 0.5) * 3.0                         # large vectors: some walks end early on |x . v| >= 6
    a0 = (rng.random((V - 1, dim)) - 0.5) * 3.0
    cfg = pyo.w2v_cfg(dim=dim, window=win, update_lr_batch=100)
    rp, ra = p0.copy(), a0.copy()
    r_lr, _ = pyo.w2v_train_slice(cfg, doc, 0, n, keep, rp, ra, paths, tab, pyo.Lcg(1), 0.025, 0, 2 * n)
    np_, na = p0.copy(), a0.copy()
    n_lr = node_major_pass(pyo, doc, keep, np_, na, paths, tab, dim, win, 0.025, cfg.init_lr, cfg.min_lr, 100, 2 * n, JB)
    assert n_lr == r_lr
    assert np.max(np.abs(rp - p0)) > 1e-3
    if JB == 1:                                                     # one pair per chunk IS the pair-major order: the same bits
        assert np.array_equal(np_, rp) and np.array_equal(na, ra)
    tol = 1e-12 * max(1.0, np.abs(rp).max(), np.abs(ra).max())
    assert np.max(np.abs(np_ - rp)) <= tol and np.max(np.abs(na - ra)) <= tol
