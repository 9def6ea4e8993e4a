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
