"""CPU tests of the torch-free one-node launcher / rendezvous (goctr_amd/launch.py) that `bench.py --gpus N` and any
N-rank host use for the control plane (RCCL unique id hand-off, barrier, max over ranks)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
from goctr_amd import launch
rank, world, local = launch.env_rank()
rdv = launch.Rendezvous(rank, world, timeout=60)
vals = rdv.allgather({"rank": rank, "sq": rank * rank})
uid = rdv.broadcast_bytes(bytes(range(128)) if rank == 0 else None)
mx = rdv.max(10.0 + rank)
rdv.barrier()
rdv.close()
if rank == 0:
    print(json.dumps({"vals": vals, "uid_ok": uid == bytes(range(128)), "max": mx, "world": world}))
if rank == 2 and os.environ.get("FAIL_RANK2"):
    sys.exit(3)
'''


def _no_gpu():
    return not os.path.exists("/dev/kfd")


@pytest.mark.parametrize("world", [1, 2, 4])
def test_rendezvous_collectives(tmp_path, world):
    from goctr_amd import launch
    script = tmp_path / "w.py"
    script.write_text(WORKER % dict(root=ROOT))
    out = tmp_path / "out.txt"
    # spawn_local passes rank 0's stdout through: capture it by running the launcher in a child interpreter
    code = ("import sys; sys.path.insert(0, %r); from goctr_amd import launch; "
            "sys.exit(launch.spawn_local(%d, [sys.executable, %r], timeout=120))" % (ROOT, world, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0, r.stderr
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["world"] == world and res["uid_ok"] and res["max"] == 10.0 + world - 1
    assert res["vals"] == [{"rank": k, "sq": k * k} for k in range(world)]
    del out, launch


def test_a_failing_rank_stops_the_job(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % dict(root=ROOT))
    code = ("import sys; sys.path.insert(0, %r); from goctr_amd import launch; "
            "sys.exit(launch.spawn_local(3, [sys.executable, %r], timeout=120))" % (ROOT, str(script)))
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}, FAIL_RANK2="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 3 and "rank 2 exited with code 3" in r.stderr


def test_bench_self_launch_fails_cleanly_without_gpu():
    """`python bench.py --gpus 2` with no launcher spawns its two ranks itself; on a GPU-less box both fail LOUDLY at
    goctr_init (no CPU fallback) and the launcher returns non-zero instead of hanging"""
    if not _no_gpu():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0
    assert "no HIP device" in r.stderr and "stopping the other ranks" in r.stderr
    assert r.stdout.strip() == ""                      # no JSON line from a job that did not run


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode == 2 and "launcher started 1 ranks" in r.stderr
