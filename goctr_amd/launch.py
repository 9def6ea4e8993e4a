"""One-node, one-process-per-GPU launcher + control-plane rendezvous for the data-parallel path.

go-ctr is single-process: there is no reference counterpart.  The DATA plane of the N-GPU run is RCCL
inside libgoctr_hip.so (goctr_comm_*); what the host needs on top is tiny -- hand the 128-byte RCCL unique id from rank 0
to the others, a barrier, a max over ranks for the timing -- and deliberately torch-free: a Unix-domain socket on rank 0
(abstract namespace, so nothing to clean up), every collective an all-gather of small JSON values through it.

Works under any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE (``python -m torch.distributed.run``: the key is derived
from MASTER_PORT + TORCHELASTIC_RUN_ID) and under ``spawn_local`` below (``python bench.py --gpus N`` with no launcher).
"""
from __future__ import annotations

import base64
import json
import os
import socket
import struct
import subprocess
import sys
import time


def _send(sock, obj):
    data = json.dumps(obj).encode()
    sock.sendall(struct.pack("<I", len(data)) + data)


def _recv(sock):
    hdr = b""
    while len(hdr) < 4:
        chunk = sock.recv(4 - len(hdr))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        hdr += chunk
    n = struct.unpack("<I", hdr)[0]
    buf = b""
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return json.loads(buf)


def rendezvous_key() -> str:
    k = os.environ.get("GOCTR_RDV_KEY")
    if k:
        return k
    return "p%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"))


class Rendezvous:
    """all ranks of ONE node; rank 0 serves.  Every collective is an all-gather of one JSON value per rank."""

    def __init__(self, rank: int, world: int, key: str | None = None, timeout: float = 300.0):
        self.rank, self.world = rank, world
        self.peers = []
        self.sock = None
        if world == 1:
            return
        addr = "\0goctr_rdv_" + (key or rendezvous_key())
        if rank == 0:
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(addr)
            srv.listen(world)
            srv.settimeout(timeout)
            peers = {}
            while len(peers) < world - 1:
                c, _ = srv.accept()
                c.settimeout(timeout)
                peers[int(_recv(c))] = c
            srv.close()
            self.peers = [peers[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(addr)
                    break
                except (ConnectionRefusedError, FileNotFoundError):
                    s.close()
                    if time.time() > deadline:
                        raise TimeoutError("rendezvous: rank 0 did not come up")
                    time.sleep(0.05)
            s.settimeout(timeout)
            _send(s, rank)
            self.sock = s

    def allgather(self, value):
        if self.world == 1:
            return [value]
        if self.rank == 0:
            vals = [value] + [_recv(p) for p in self.peers]
            for p in self.peers:
                _send(p, vals)
            return vals
        _send(self.sock, value)
        return _recv(self.sock)

    def barrier(self):
        self.allgather(0)

    def max(self, x: float) -> float:
        return max(self.allgather(float(x)))

    def broadcast_bytes(self, data: bytes | None) -> bytes:
        vals = self.allgather(base64.b64encode(data).decode() if self.rank == 0 else None)
        return base64.b64decode(vals[0])

    def close(self):
        for p in self.peers:
            p.close()
        if self.sock:
            self.sock.close()
        self.peers, self.sock = [], None


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))


def init_comm(rdv: Rendezvous, local_rank: int):
    """bind this process to its GPU and, for world > 1, create the RCCL communicator: rank 0 draws the unique id
    (goctr_comm_unique_id), the rendezvous distributes it, every rank calls goctr_comm_init.  Returns the RCCL world size
    the library reports back."""
    import ctypes as C

    from . import capi
    L = capi.init(local_rank)
    if rdv.world > 1:
        idbuf = (C.c_uint8 * 128)()
        if rdv.rank == 0:
            capi.check(L.goctr_comm_unique_id(idbuf))
        raw = rdv.broadcast_bytes(bytes(idbuf) if rdv.rank == 0 else None)
        idbuf = (C.c_uint8 * 128)(*raw)
        capi.check(L.goctr_comm_init(C.c_int(rdv.rank), C.c_int(rdv.world), idbuf))
    r, w = C.c_int(0), C.c_int(0)
    capi.check(L.goctr_comm_world(C.byref(r), C.byref(w)))
    return w.value


def spawn_local(n: int, argv: list[str], timeout: float | None = None) -> int:
    """run ``argv`` as n ranks on this node (RANK = LOCAL_RANK = 0..n-1), rank 0's stdout passed through.  If any rank
    exits non-zero the others are terminated (by pid) so that nobody waits on a dead peer.  Returns the exit code."""
    key = "self%d_%d" % (os.getpid(), int(time.time() * 1000) % 1_000_000)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), GOCTR_RDV_KEY=key,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen(argv, env=env, stdout=out))
    deadline = time.time() + timeout if timeout else None
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"launch: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
        if (rc != 0 or (deadline and time.time() > deadline)) and live:
            if rc == 0:
                rc = 124
                print("launch: timeout; stopping all ranks", file=sys.stderr)
            for r in live:
                procs[r].terminate()
            for r in list(live):
                try:
                    procs[r].wait(10)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
                live.discard(r)
        time.sleep(0.02)
    return rc
