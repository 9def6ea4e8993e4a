"""world_size-2 gloo test (CPU) of the data-parallel scheme libgoctr_hip.so uses across GPUs:
rows are sharded over ranks, every rank runs forward/backward on its shard with the GLOBAL batch size in
the 1/B factors, the flat gradient buffer plus the BCE sum travel in ONE all-reduce(sum), then every rank
applies the identical Adam step.  The oracle stands in for the device kernels here (no GPU in this
container); the test pins the math of the exchange step: sharded == single-process full batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

U, T, D, C_, B = 5, 3, 7, 5, 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    rng = np.random.default_rng(0)
    X = rng.random((B, U + T * D + D + C_), dtype=np.float32)
    Y = (rng.random(B) < 0.5).astype(np.float32)
    return X, Y


def _model(pyoracle):
    m = pyoracle.CtrModel(pyoracle.DIN, U, T, D, C_)
    rng = np.random.default_rng(1)
    for w in (m.W0, m.W1, m.W2):
        w[:] = (rng.standard_normal(w.shape) * 0.2).astype(np.float32)
    return m


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle
    X, Y = _data()
    m = _model(pyoracle)
    nloc = B // world
    Xs, Ys = X[rank * nloc:(rank + 1) * nloc], Y[rank * nloc:(rank + 1) * nloc]
    cost, g, _ = m.loss_grad(Xs, Ys)                       # local mean over nloc rows
    # device convention: un-normalised local sums scaled by 1/B_global  ==  local mean * nloc/B
    flat = np.concatenate([g["W0"].ravel(), g["W1"].ravel(), g["W2"].ravel(), g["att0"].ravel(),
                           [cost]]).astype(np.float32) * (nloc / B)
    t = torch.from_numpy(flat)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)               # ONE collective per step
    if rank == 0:
        out.put(t.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_step_equals_full_batch(oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    red = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    X, Y = _data()
    m = _model(oracle)
    cost, g, _ = m.loss_grad(X, Y)
    full = np.concatenate([g["W0"].ravel(), g["W1"].ravel(), g["W2"].ravel(), g["att0"].ravel(), [cost]])
    assert np.max(np.abs(red - full)) <= 1e-6 + 1e-4 * np.max(np.abs(full))   # fp32 summation order only
    assert abs(red[-1] - cost) < 1e-6


def test_rank_row_offsets_make_the_dropout_mask_global(oracle):
    """the hash mask is keyed on the GLOBAL row (rank*B_local + row): two shards reproduce the mask of the
    unsharded batch"""
    import ctypes as C
    L = oracle.lib()
    full = np.array([[L.orc_dropout_keep(9, 2, 0, r, c, C.c_float(0.3)) for c in range(16)] for r in range(8)])
    for rank in range(2):
        shard = np.array([[L.orc_dropout_keep(9, 2, 0, rank * 4 + r, c, C.c_float(0.3)) for c in range(16)]
                          for r in range(4)])
        assert np.array_equal(shard, full[rank * 4:(rank + 1) * 4])


# ---------------------------------------------------------------- item2vec: exchange of parameter deltas
def _w2v_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle
    rng = np.random.default_rng(7)
    V, n, dim = 40, 3000, 8
    doc = rng.integers(0, V, size=n).astype(np.int32)
    counts = np.bincount(doc, minlength=V) + 1
    p0 = (rng.random((V, dim)) - 0.5) / dim
    cfg = pyoracle.w2v_cfg(dim=dim, optimizer="hs")
    paths = pyoracle.huffman_paths(counts)
    # every rank: full replica, its own contiguous corpus shard (IndexPerThread-style split, modelutil.go:32-41)
    lo, hi = rank * n // world, (rank + 1) * n // world
    param, aux = p0.copy(), np.zeros((V - 1, dim))
    pyoracle.w2v_train_slice(cfg, doc, lo, hi, None, param, aux, paths, pyoracle.sigmoid_table(), pyoracle.Lcg(1 + rank), 0.025, 0, n)
    # the exchange libgoctr_hip.so does after a pass (csrc/w2v.hip exchange_deltas): p = p0 + sum_r (p_r - p0)
    dparam, daux = torch.from_numpy(param - p0), torch.from_numpy(aux.copy())
    dist.all_reduce(dparam); dist.all_reduce(daux)
    merged_p, merged_a = p0 + dparam.numpy(), daux.numpy()
    if rank == 0:
        # reference for the math: both shards trained from the same start, deltas added up in one process
        exp_p, exp_a = p0.copy(), np.zeros((V - 1, dim))
        for r in range(world):
            pr, ar = p0.copy(), np.zeros((V - 1, dim))
            l2, h2 = r * n // world, (r + 1) * n // world
            pyoracle.w2v_train_slice(cfg, doc, l2, h2, None, pr, ar, paths, pyoracle.sigmoid_table(), pyoracle.Lcg(1 + r), 0.025, 0, n)
            exp_p += pr - p0
            exp_a += ar
        np.save(out, np.array([np.max(np.abs(merged_p - exp_p)), np.max(np.abs(merged_a - exp_a)),
                               np.max(np.abs(merged_p - p0))]))
    dist.barrier()
    dist.destroy_process_group()


def test_item2vec_delta_exchange_world2(tmp_path):
    out = str(tmp_path / "w2v_dp.npy")
    mp.spawn(_w2v_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ep, ea, moved = np.load(out)
    assert ep <= 1e-15 and ea <= 1e-15      # all-reduce(sum) of the deltas == adding the shards' deltas in one process
    assert moved > 1e-4                     # and the replicas really moved


def _w2v_avg_worker(rank, world, port, out):
    """the HOGWILD passes' exchange (round 5; csrc/w2v.hip exchange_deltas(avg = true)): a pass is cut into segments, after each
    the ranks all-reduce their parameter deltas since the common snapshot AND a per-row 'this rank changed the row' flag, and
    every row becomes snapshot + sum of deltas / number of ranks that changed it"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle
    rng = np.random.default_rng(8)
    V, n, dim, nseg = 60, 4000, 8, 3
    doc = np.concatenate([rng.integers(0, 30, size=n // 2), rng.integers(20, V, size=n // 2)]).astype(np.int32)   # rank 0 never sees words >= 30
    counts = np.bincount(doc, minlength=V) + 1
    p0 = (rng.random((V, dim)) - 0.5) / dim
    cfg = pyoracle.w2v_cfg(dim=dim, optimizer="hs")
    paths = pyoracle.huffman_paths(counts)
    sig = pyoracle.sigmoid_table()

    def run(ranks, reduce):
        P, A = p0.copy(), np.zeros((V - 1, dim))
        lcg = {r: pyoracle.Lcg(1 + r) for r in ranks}
        lr = {r: 0.025 for r in ranks}
        for s_ in range(nseg):
            ds = {}
            for r in ranks:
                lo, hi = r * n // world, (r + 1) * n // world
                a0, a1 = lo + (hi - lo) * s_ // nseg, lo + (hi - lo) * (s_ + 1) // nseg
                p, a = P.copy(), A.copy()
                lr[r], _ = pyoracle.w2v_train_slice(cfg, doc[a0:a1], 0, a1 - a0, None, p, a, paths, sig, lcg[r], lr[r], world * (a0 - lo), n)
                ds[r] = (p - P, a - A)
            for M, k in ((P, 0), (A, 1)):
                d = sum(ds[r][k] for r in ranks)
                cnt = sum((np.abs(ds[r][k]).max(1) > 0).astype(np.float64) for r in ranks)
                d, cnt = reduce(d, cnt)
                M += d / np.maximum(cnt, 1.0)[:, None]
        return P, A

    def allreduce(d, cnt):
        td, tc = torch.from_numpy(d.copy()), torch.from_numpy(cnt.copy())
        dist.all_reduce(td); dist.all_reduce(tc)
        return td.numpy(), tc.numpy()

    P, A = run([rank], allreduce)                       # this rank's shard, the other rank's through the collectives
    if rank == 0:
        eP, eA = run(list(range(world)), lambda d, c: (d, c))      # both shards in one process
        # a row only ONE rank trained keeps its whole update: words >= 30 occur in rank 1's half only
        solo = np.max(np.abs(P[40:] - p0[40:]))
        np.save(out, np.array([np.max(np.abs(P - eP)), np.max(np.abs(A - eA)), solo]))
    dist.barrier()
    dist.destroy_process_group()


def test_item2vec_hogwild_exchange_rule_world2(tmp_path):
    out = str(tmp_path / "w2v_avg.npy")
    mp.spawn(_w2v_avg_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ep, ea, solo = np.load(out)
    assert ep <= 1e-15 and ea <= 1e-15      # segments + all-reduce of (delta, touched) == the rule evaluated in one process
    assert solo > 1e-4                      # rows of one rank's shard moved by their full delta


# ---------------------------------------------------------------- sklearn-port MLP: sharded rows == full batch
def _mlp_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle
    rng = np.random.default_rng(5)
    F, H, n = 9, 6, 48
    X = rng.random((n, F)); Y = (rng.random((n, 1)) < 0.5).astype(np.float64)
    cfg = pyoracle.mlp_cfg([F, H, 1], "relu", alpha=1e-3)
    theta = rng.standard_normal(pyoracle.mlp_nparams(cfg)) * 0.3
    nloc = n // world
    Xs, Ys = X[rank * nloc:(rank + 1) * nloc], Y[rank * nloc:(rank + 1) * nloc]
    alpha = 1e-3
    cfg0 = pyoracle.mlp_cfg([F, H, 1], "relu", alpha=0.0)
    _, d_loc = pyoracle.mlp_loss_grad(cfg0, theta.copy(), Xs, Ys)        # data term only: s_r / nloc
    coef = np.zeros(theta.size, bool)                                    # packed order per layer: [intercepts | coefs]
    coef[H:H + F * H] = True
    coef[H + F * H + 1:] = True
    # what csrc/mlp.hip exchanges: the rank's slab sums with the GLOBAL 1/N plus its 1/world share of the penalty gradient
    t = torch.from_numpy(d_loc * (nloc / n) + np.where(coef, (alpha / n) * theta / world, 0.0))
    dist.all_reduce(t)
    if rank == 0:
        loss_full, g_full = pyoracle.mlp_loss_grad(cfg, theta.copy(), X, Y)
        np.save(out, np.array([np.max(np.abs(t.numpy() - g_full)), np.max(np.abs(g_full))]))
    dist.barrier()
    dist.destroy_process_group()


def test_mlp_sharded_gradient_world2(tmp_path):
    out = str(tmp_path / "mlp_dp.npy")
    mp.spawn(_mlp_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    err, scale = np.load(out)
    assert scale > 1e-3 and err <= 1e-14 * max(1.0, scale)


# ---------------------------------------------------------------- trainable-embedding extension, world 2
def _emb_case():
    rng = np.random.default_rng(21)
    Bg, Ue, Te, De, Ce, V = 8, 5, 6, 8, 4, 19
    ub = rng.integers(-1, V, size=(Bg, Te)).astype(np.int32)
    items = rng.integers(0, V, size=Bg).astype(np.int32)
    uf = rng.random((Bg, Ue), dtype=np.float32)
    cf = rng.random((Bg, Ce), dtype=np.float32)
    y = (rng.random(Bg) < 0.5).astype(np.float32)
    E = rng.standard_normal((V, De)) * 0.5
    return Bg, Ue, Te, De, Ce, V, ub, items, uf, cf, y, E


def _emb_model(pyoracle, Ue, Te, De, Ce):
    m = pyoracle.CtrModel(pyoracle.DIN, Ue, Te, De, Ce, H1=16, H2=8)
    rng = np.random.default_rng(2)
    for w in (m.W0, m.W1, m.W2):
        w[:] = (rng.standard_normal(w.shape) * 0.3).astype(np.float32)
    return m


def _emb_worker(rank, world, port, out, fixed=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle
    Bg, Ue, Te, De, Ce, V, ub, items, uf, cf, y, E = _emb_case()
    m = _emb_model(pyoracle, Ue, Te, De, Ce)
    n = Bg // world
    sl = slice(rank * n, (rank + 1) * n)
    W, r = world, rank
    # The device's bucketed exchange (csrc/emb_train.h + ctr.hip launch_emb_exchange), step by step in numpy:
    #  0. local row gradients scaled by 1/B_global, 2^-44 fixed point, one accumulator row per touched id
    _, dE = m.emb_loss_grad(E, ub[sl], items[sl], uf[sl], cf[sl], y[sl])             # mean over n local rows
    dE = dE * (n / Bg)
    ids_b = np.concatenate([ub[sl].ravel(), items[sl]])
    touched = np.unique(ids_b[(ids_b >= 0) & (ids_b < V)])
    #  1. owner-major numbering: pidx(id) = (id % W) * Vw + id / W; slots = touched ids in ascending pidx order, so the
    #     rows for owner o are one contiguous range
    Vw = -(-(-(-V // W)) // 4) * 4
    pidx = (touched % W) * Vw + touched // W
    order = np.argsort(pidx, kind="stable")
    slot_id = touched[order]
    accum = np.rint(dE[slot_id] * 2.0 ** 44).astype(np.int64)
    off = np.searchsorted(slot_id % W, np.arange(W + 1), side="left")             # emb_bucket_bounds_kernel
    cnt = np.diff(off).astype(np.int32)
    assert np.all(np.diff(slot_id % W) >= 0) and all(np.all(np.diff(slot_id[off[o]:off[o + 1]]) > 0) for o in range(W))
    if fixed:
        # Round 3, the graph-capturable form (csrc/emb_train.h, end; ctr.hip emb_exchange_*): every bucket padded to S = the
        # largest bucket of any rank (ids -1, rows 0), the owners' lists to R = min(Vw, W S): uniform transfers whose sizes
        # the host knows beforehand; the counts travel in-band as the padding
        smax = torch.tensor([int(cnt.max()) if cnt.size else 0], dtype=torch.int32)
        dist.all_reduce(smax, op=dist.ReduceOp.MAX)                                  # (on the device: once, when the plan is built)
        S = max(4, -(-int(smax.item()) // 4) * 4)
        R = min(Vw, W * S)
        sid = np.full((W, S), -1, np.int32); srow = np.zeros((W, S, De), np.int64)
        for o in range(W):                                                           # emb_pack_send_kernel
            k = off[o + 1] - off[o]
            sid[o, :k] = slot_id[off[o]:off[o + 1]]; srow[o, :k] = accum[off[o]:off[o + 1]]
        rid = np.full((W, S), -1, np.int32); rrow = np.zeros((W, S, De), np.int64)
        reqs, bufs = [], []
        for p in range(W):                                                           # uniform all-to-all: S entries per peer
            if p == r:
                rid[r] = sid[r]; rrow[r] = srow[r]
                continue
            ti, tr = torch.from_numpy(sid[p].copy()), torch.from_numpy(srow[p].copy())
            ri, rr = torch.zeros(S, dtype=torch.int32), torch.zeros((S, De), dtype=torch.int64)
            bufs.append((p, ri, rr))
            reqs += [dist.isend(ti, p, tag=1), dist.isend(tr, p, tag=2), dist.irecv(ri, p, tag=1), dist.irecv(rr, p, tag=2)]
        for q in reqs:
            q.wait()
        for p, ri, rr in bufs:
            rid[p] = ri.numpy(); rrow[p] = rr.numpy()
        rids = rid.ravel().astype(np.int64); rrows = rrow.reshape(-1, De)
        live = rids >= 0                                                             # the owner kernels skip the padding
        assert np.all(rids[live] % W == r)
        red_ids = np.unique(rids[live])
        assert red_ids.size <= R
        red = np.zeros((red_ids.size, De), np.int64)
        np.add.at(red, np.searchsorted(red_ids, rids[live]), rrows[live])
        lr = np.float32(0.05)
        delta = (lr * (red.astype(np.float64) * 2.0 ** -44).astype(np.float32)).astype(np.float32)
        gid = np.full(R, -1, np.int32); gdl = np.zeros((R, De), np.float32)          # emb_pad_ids_kernel
        gid[:red_ids.size] = red_ids; gdl[:red_ids.size] = delta
        all_ids = [torch.zeros(R, dtype=torch.int32) for _ in range(W)]
        all_dl = [torch.zeros((R, De), dtype=torch.float32) for _ in range(W)]
        dist.all_gather(all_ids, torch.from_numpy(gid))                              # fixed-size all-gathers
        dist.all_gather(all_dl, torch.from_numpy(gdl))
        g_ids = np.concatenate([t.numpy() for t in all_ids]); g_delta = np.concatenate([t.numpy() for t in all_dl])
        keep = g_ids >= 0                                                            # emb_apply_gathered_kernel skips -1
        assert np.unique(g_ids[keep]).size == int(keep.sum())
        E32 = E.astype(np.float32)
        new = E32.copy()
        new[g_ids[keep]] = E32[g_ids[keep]] - g_delta[keep]
        sent = float(W * S * (4 + 8 * De) + W * R * (4 + 4 * De))
        out.put((rank, new, float(lr), sent))
        dist.barrier()
        dist.destroy_process_group()
        return
    #  2. counts all-gather, then all-to-all-v of (ids, rows) to the owners
    allcnt = [torch.zeros(W, dtype=torch.int32) for _ in range(W)]
    dist.all_gather(allcnt, torch.from_numpy(cnt))
    allcnt = np.stack([t.numpy() for t in allcnt])                                   # [src][owner]
    send_ids = [torch.from_numpy(slot_id[off[o]:off[o + 1]].astype(np.int32).copy()) for o in range(W)]
    send_rows = [torch.from_numpy(accum[off[o]:off[o + 1]].copy()) for o in range(W)]
    recv_ids = [torch.zeros(int(allcnt[s_, r]), dtype=torch.int32) for s_ in range(W)]
    recv_rows = [torch.zeros((int(allcnt[s_, r]), De), dtype=torch.int64) for s_ in range(W)]
    reqs = []
    for p in range(W):                                    # gloo has no all_to_all: pairwise isend / irecv (self: copy)
        if p == r:
            recv_ids[r].copy_(send_ids[r]); recv_rows[r].copy_(send_rows[r])
            continue
        reqs += [dist.isend(send_ids[p], p, tag=1), dist.isend(send_rows[p], p, tag=2),
                 dist.irecv(recv_ids[p], p, tag=1), dist.irecv(recv_rows[p], p, tag=2)]
    for q in reqs:
        q.wait()
    rids = np.concatenate([t.numpy() for t in recv_ids]).astype(np.int64)
    rrows = np.concatenate([t.numpy() for t in recv_rows]) if rids.size else np.zeros((0, De), np.int64)
    assert np.all(rids % W == r)
    #  3. owner: unique ids of my bucket in ascending order (the bucket scan), exact integer sums, delta = lr * float(sum)
    red_ids = np.unique(rids)
    red = np.zeros((red_ids.size, De), np.int64)
    np.add.at(red, np.searchsorted(red_ids, rids), rrows)
    lr = np.float32(0.05)
    delta = (lr * (red.astype(np.float64) * 2.0 ** -44).astype(np.float32)).astype(np.float32)
    #  4. all-gather of (ids, deltas); every replica applies every delta
    nred = [torch.zeros(1, dtype=torch.int32) for _ in range(W)]
    dist.all_gather(nred, torch.tensor([red_ids.size], dtype=torch.int32))
    nred = [int(t.item()) for t in nred]
    g_ids, g_delta = [], []
    for s_ in range(W):
        ti = torch.from_numpy(red_ids.astype(np.int32).copy()) if s_ == r else torch.zeros(nred[s_], dtype=torch.int32)
        td = torch.from_numpy(delta.copy()) if s_ == r else torch.zeros((nred[s_], De), dtype=torch.float32)
        dist.broadcast(ti, src=s_)
        dist.broadcast(td, src=s_)
        g_ids.append(ti.numpy()); g_delta.append(td.numpy())
    g_ids = np.concatenate(g_ids); g_delta = np.concatenate(g_delta)
    assert np.unique(g_ids).size == g_ids.size               # every id has exactly one owner
    E32 = E.astype(np.float32)
    new = E32.copy()
    new[g_ids] = E32[g_ids] - g_delta
    sent = float(sum(int(cnt[o]) * (4 + 8 * De) for o in range(W)) + W * red_ids.size * (4 + 4 * De))
    out.put((rank, new, float(lr), sent))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fixed", [False, True])
@pytest.mark.parametrize("world", [2, 4])
def test_embedding_sparse_exchange_bucketed(oracle, world, fixed):
    """owner = id % world, all-to-all of the deduplicated (id, fixed-point row) pairs, exact
    owner-side sums, all-gather of (id, delta): every replica ends up with the SAME bits, equal to one SGD step on the
    full batch's oracle gradient up to the 2^-44 rounding of each rank's contribution"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_emb_worker, args=(r, world, port, q, fixed)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    Bg, Ue, Te, De, Ce, V, ub, items, uf, cf, y, E = _emb_case()
    _, full = _emb_model(oracle, Ue, Te, De, Ce).emb_loss_grad(E, ub, items, uf, cf, y)
    assert np.abs(full).max() > 1e-4
    tables = {r: t for r, t, _, _ in got}
    lr = got[0][2]
    for r in range(1, world):
        assert np.array_equal(tables[0], tables[r])                                   # replicas bit-identical
    want = E.astype(np.float32).astype(np.float64) - lr * full
    # per element: world contributions rounded to 2^-44 each, one float32 rounding of the delta, one of the subtraction
    assert np.abs(tables[0] - want).max() <= lr * (world * 2.0 ** -44 + 2.0 ** -23 * np.abs(full).max()) + 2.0 ** -23 * np.abs(want).max()
    touched = np.unique(np.concatenate([ub.ravel(), items]))
    touched = touched[touched >= 0]
    untouched = np.setdiff1d(np.arange(V), touched)
    assert np.array_equal(tables[0][untouched], E.astype(np.float32)[untouched])     # untouched rows keep their bits
    assert all(s > 0 for _, _, _, s in got)


# ---- the FIXED bucket sizes over a whole plan: S and R are drawn once, when the plan is built, from every batch of
# every rank; each later step only packs into them.  World 2, several batches, a hot id owned by rank 1, a batch whose bucket for one
# owner is EMPTY -- and a batch that is NOT in the plan, whose bucket must be refused instead of silently truncated.
def _plan_case():
    rng = np.random.default_rng(33)
    V, T, nb, Bg = 41, 5, 6, 8
    ub = ((rng.zipf(1.3, size=(nb, Bg, T)) - 1) % V).astype(np.int32)
    ub[rng.random(ub.shape) < 0.2] = -1
    items = ((rng.zipf(1.3, size=(nb, Bg)) - 1) % V).astype(np.int32)
    ub[:, :, 0] = 7                                   # the hot id: owner 7 % 2 = rank 1
    ub[3] = np.where(ub[3] % 2 == 1, ub[3] - 1, ub[3])   # batch 3: only even ids ...
    ub[3][ub[3] < 0] = -1
    items[3] -= items[3] % 2                          # ... so owner 1's bucket is EMPTY for batch 3
    return V, T, nb, Bg, ub, items


def _buckets(ids, V, W):
    """slots of one rank's batch: touched ids in owner-major order (pidx = (id % W) * Vw + id / W) and the bucket bounds"""
    Vw = -(-(-(-V // W)) // 4) * 4
    t = np.unique(ids[(ids >= 0) & (ids < V)])
    slot_id = t[np.argsort((t % W) * Vw + t // W, kind="stable")]
    off = np.searchsorted(slot_id % W, np.arange(W + 1), side="left")
    return slot_id, off, Vw


def _pack(slot_id, off, W, S):
    """emb_pack_send_kernel's contract: S entries per owner, -1 padded; a bucket longer than S is an error, never a truncation"""
    sid = np.full((W, S), -1, np.int32)
    for o in range(W):
        k = off[o + 1] - off[o]
        if k > S:
            raise OverflowError(f"bucket for owner {o} holds {k} ids, the plan's bound is {S}")
        sid[o, :k] = slot_id[off[o]:off[o + 1]]
    return sid


def _plan_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V, T, nb, Bg, ub, items = _plan_case()
    W, n = world, Bg // world
    sl = slice(rank * n, (rank + 1) * n)
    per_batch = [_buckets(np.concatenate([ub[k, sl].ravel(), items[k, sl]]), V, W) for k in range(nb)]
    # plan build: the bound is the largest bucket of ANY batch on ANY rank (ctr.hip: one all-gather of the local maxima)
    smax = torch.tensor([max(int(np.diff(off).max()) for _, off, _ in per_batch)], dtype=torch.int32)
    dist.all_reduce(smax, op=dist.ReduceOp.MAX)
    S = max(4, -(-int(smax.item()) // 4) * 4)
    Vw = per_batch[0][2]
    R = min(Vw, W * S)
    seen_empty, worst_owner = False, 0
    for k in range(nb):
        slot_id, off, _ = per_batch[k]
        sid = _pack(slot_id, off, W, S)                                  # never raises for a batch of the plan
        seen_empty |= bool(np.any(np.diff(off) == 0))
        rid = np.full((W, S), -1, np.int32)
        reqs, bufs = [], []
        for p in range(W):
            if p == rank:
                rid[rank] = sid[rank]
                continue
            ri = torch.zeros(S, dtype=torch.int32)
            bufs.append((p, ri))
            reqs += [dist.isend(torch.from_numpy(sid[p].copy()), p, tag=k), dist.irecv(ri, p, tag=k)]
        for q in reqs:
            q.wait()
        for p, ri in bufs:
            rid[p] = ri.numpy()
        live = rid[rid >= 0]
        assert np.all(live % W == rank)                                  # only ids this rank owns arrive
        n_red = np.unique(live).size
        assert n_red <= R, (n_red, R)                                    # the owner-side list fits its fixed size
        worst_owner = max(worst_owner, n_red)
    # a batch that was not in the plan: rank 0 suddenly touches more ids of one owner than any planned batch did
    refused = False
    try:
        rogue = np.arange(1, 2 * (S + 2), 2, dtype=np.int32) % V        # S + 2 distinct odd ids: owner 1
        sid_, off_, _ = _buckets(rogue, V, W)
        _pack(sid_, off_, W, S)
    except OverflowError:
        refused = True
    out.put((rank, S, R, seen_empty, worst_owner, refused))
    dist.barrier()
    dist.destroy_process_group()


def test_embedding_fixed_buckets_hold_every_batch_of_the_plan():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] and got[0][2] == got[1][2]             # both ranks drew the same S and R
    assert any(g[3] for g in got)                                        # an empty bucket was exercised
    assert all(g[5] for g in got)                                        # the unplanned batch was refused on both ranks
    V, T, nb, Bg, ub, items = _plan_case()
    assert got[0][1] >= 4 and got[0][2] <= -(-(-(-V // 2)) // 4) * 4     # R never exceeds the owner's share of the vocabulary
