"""8-GPU readiness that can be proven without 8 GPUs: `python bench.py --gpus 8` end to end against a STUB of the C-ABI.

The stub (generated below from goctr_amd.capi.SYMBOLS, built with gcc) exports every symbol include/goctr.h declares and
computes nothing; GOCTR_LIB points goctr_amd.capi at it.  What runs for real is everything AROUND the library in an
N-rank job: bench.py spawning its own ranks (goctr_amd/launch.py), the Unix-socket rendezvous, the RCCL unique id drawn on
rank 0 and handed to every rank (the stub's goctr_comm_init rejects any other 128 bytes), the barrier-bracketed timed
region with the max over ranks, the per-rank JSON merge (rccl_world, per_rank_ms_per_step,
sparse_exchange_bytes_per_step_per_rank) and rank 0 printing exactly ONE line -- and the refusal of a run whose
communicator is smaller than --gpus.  Numbers on the line are meaningless here (the stub sleeps); the scaling curve itself
stays unmeasured until the driver has an 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SPECIAL = r'''
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static int g_rank = 0, g_world = 1;
static const char* g_err = "stub";
const char* goctr_last_error(void) { return g_err; }
const char* goctr_version(void) { return "goctr stub (no device, no arithmetic)"; }
int goctr_comm_unique_id(uint8_t* id) { for (int i = 0; i < 128; ++i) id[i] = (uint8_t)(i * 7 + 3); return 0; }
int goctr_comm_init(int rank, int world, const uint8_t* id) {
  for (int i = 0; i < 128; ++i) if (id[i] != (uint8_t)(i * 7 + 3)) { g_err = "stub: rank got a different unique id than rank 0 drew"; return -1; }
  g_rank = rank; g_world = world; return 0;
}
int goctr_comm_world(int* r, int* w) { *r = g_rank; *w = g_world - (getenv("STUB_WORLD_LIE") ? 1 : 0); return 0; }
static const char* kNames[17] = {"attn_fwd", "gemm_fwd0", "gemm_fwd1", "gemm_out", "bwd_dz1", "bwd_dz0", "bwd_dp", "attn_bwd", "dW0", "dW1",
                                 "dW2", "reduce", "allreduce", "adam", "chain", "emb_train", "emb_grad"};
const char* goctr_prof_name(int id) { return id >= 0 && id < 17 ? kNames[id] : "?"; }
const char* goctr_prof_kernel(int id) { (void)id; return ""; }
int goctr_prof_get(int id, double* ms, int64_t* n) {
  const int on = id == 0 || id == 7 || id == 8 || id == 11 || id == 12 || id == 14;
  *ms = on ? 0.2 * (id + 1) : 0.0; *n = on ? 10 : 0; return 0;
}
int goctr_train_steps(void* m, void* e, void* d, const void* cfg, int64_t first, int n, float* costs) {
  (void)m; (void)e; (void)d; (void)cfg; (void)first; (void)costs;
  usleep((useconds_t)(300 * n * (1 + g_rank)));        /* later ranks are slower: the line must carry the MAX */
  return 0;
}
int goctr_model_sparse_exchange_bytes(void* m, double* b) { (void)m; *b = 1000.0 * (g_rank + 1); return 0; }
size_t goctr_mlp_nparams(const void* p) { (void)p; return 0; }
'''
SPECIAL_NAMES = {"goctr_last_error", "goctr_version", "goctr_comm_unique_id", "goctr_comm_init", "goctr_comm_world", "goctr_prof_name",
                 "goctr_prof_kernel", "goctr_prof_get", "goctr_train_steps", "goctr_model_sparse_exchange_bytes", "goctr_mlp_nparams"}


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    sys.path.insert(0, ROOT)
    from goctr_amd import capi
    d = tmp_path_factory.mktemp("stub")
    src = d / "stub.c"
    body = SPECIAL + "".join(f"int {s}() {{ return 0; }}\n" for s in capi.SYMBOLS if s not in SPECIAL_NAMES)
    src.write_text(body)
    so = d / "libgoctr_stub.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-w", str(src), "-o", str(so)], check=True)
    return str(so)


def _run(stub, extra, env_extra=None, gpus=8):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(GOCTR_LIB=stub, GOCTR_BENCH_TIMEOUT="120")
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "6", "--warmup", "2", "--rows",
                           "16384", "--no-cpu-baseline", "--no-serving"] + extra, capture_output=True, text=True, timeout=300, env=env)


def test_bench_gpus8_runs_end_to_end_against_the_stub(stub):
    r = _run(stub, [])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["rccl_world"] == 8 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    per = d["per_rank_ms_per_step"]
    assert len(per) == 8 and all(p > 0 for p in per)
    # (the timed region ends with a barrier, so every rank's figure includes the wait for the slowest -- the stub's rank 7)
    assert abs(d["ms_per_step"] - max(per)) <= 0.02 * max(per) + 1e-3    # the line carries the max over ranks
    assert d["config"]["global_batch"] == 8 * 8192 and d["config"]["parallelism"] == "dp8"
    assert d["value"] == pytest.approx(8 * 8192 / (d["ms_per_step"] * 1e-3), rel=0.02)
    # the headline is the MEDIAN of 9 back-to-back regions of exactly K steps; all of them are on the line
    reg = d["timed_regions_ms"]
    assert d["timed_regions"] == 9 and len(reg) == 9 and all(x > 0 for x in reg)
    assert d["ms_per_step"] * d["steps"] == pytest.approx(sorted(reg)[4], rel=1e-3)
    assert d["timed_region_min_ms"] == min(reg) and d["timed_region_max_ms"] == max(reg)
    assert "cpu_baseline" not in d                         # rank 0 at N = 1 only
    assert d["roofline"]["kernel"] in ("chain", "dW0", "attn_fwd", "attn_bwd", "reduce")
    assert d["replicas_bit_identical"] is True            # (every rank checksums its weights; the stub leaves them zero)


def test_bench_gpus8_train_emb_line_carries_the_exchange_bytes(stub):
    r = _run(stub, ["--train-emb", "0.01", "--no-roofline"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][0])
    assert d["config"]["train_embeddings"] is True and d["rccl_world"] == 8
    assert d["sparse_exchange_bytes_per_step_per_rank"] == 1000.0      # rank 0's own figure
    assert d["replicas_bit_identical"] is True            # (+ the first 65 536 table rows)


def test_bench_fails_when_the_communicator_is_smaller_than_gpus(stub):
    r = _run(stub, ["--no-roofline"], {"STUB_WORLD_LIE": "1"}, gpus=4)
    assert r.returncode != 0
    assert "RCCL communicator has 3 ranks" in r.stderr
    assert not [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]


def test_bench_under_an_external_launcher_env(stub):
    """the driver's form: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` sets RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_PORT itself; emulate two such ranks by hand"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(GOCTR_LIB=stub, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(20000 + os.getpid() % 20000), TORCHELASTIC_RUN_ID="t")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--rows", "16384", "--no-cpu-baseline",
           "--no-serving", "--no-roofline"]
    ps = [subprocess.Popen(cmd, env=dict(env, RANK=str(k), LOCAL_RANK=str(k)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
          for k in range(2)]
    outs = [p.communicate(timeout=240) for p in ps]
    assert all(p.returncode == 0 for p in ps), [o[1][-800:] for o in outs]
    assert outs[1][0].strip() == ""                        # only rank 0 prints
    d = json.loads(outs[0][0].strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["rccl_world"] == 2 and len(d["per_rank_ms_per_step"]) == 2


def test_flop_constants_match_the_surveys_per_sample_figures():
    """the numerators of every `roofline.frac` bench.py prints for the three dense models (the
    sklearn-port MLP line priced three F x H GEMMs per step where the model has two -- no input gradient)."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.mlp_flops_per_sample(281, 100) == 113_000                          # cfg2: fwd 56 400 + bwd 56 600
    assert bench.ctr_flops_per_sample("youtube", 52, 50, 64, 53) == 282_880          # cfg4 MLP stage: fwd 125 360 + bwd 157 520
    # cfg3 MLP stage 212 480 + what SURVEY books under the attention stage: att0's weight gradient (2 T) and the T x D dots of its backward
    assert bench.ctr_flops_per_sample("din", 52, 50, 16, 53) == 212_480 + 2 * 50 + 2 * 50 * 16
    # ... and the per-launch figures the DIN / YouTube lines divide by a launch's duration add up to exactly that per step
    for kind, D, B in (("din", 16, 8192), ("youtube", 64, 16384)):
        saved = dict(bench.CFG)
        try:
            bench.CFG.update(D=D, B=B, KIND=kind)
            w = bench.kernel_work()
            per_step = w["chain"][1] + w["dW0"][1]
            assert per_step == pytest.approx(B * bench.ctr_flops_per_sample(kind, 52, 50, D, 53), rel=1e-12)
        finally:
            bench.CFG.clear(); bench.CFG.update(saved)


def test_bench_strong_scaling_splits_baselines_global_batch(stub):
    """--strong (the GLOBAL batch of 8192 is what shards): every rank steps 8192 / N rows"""
    r = _run(stub, ["--strong", "--no-roofline"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][0])
    assert d["scaling"] == "strong" and d["n_gpus"] == 8
    assert d["config"]["global_batch"] == 8192 and "1024 rows per GPU" in d["config"]["workload"]
    assert d["value"] == pytest.approx(8192 / (d["ms_per_step"] * 1e-3), rel=0.02)
    assert d["without_preload"] is None                   # N > 1 never pre-loads
    r = _run(stub, ["--strong", "--no-roofline"], gpus=3)
    assert r.returncode == 2 and "multiple of --gpus" in r.stderr


def test_bench_n1_line_carries_the_figure_without_the_preload(stub):
    """only N = 1 runs pre-load the GPU with a scratch model's training steps; the N = 1 line therefore also
    carries the same measurement taken BEFORE that pre-load -- the protocol of every N > 1 line"""
    r = _run(stub, ["--no-roofline"], {"GOCTR_BENCH_PRELOAD_STEPS": "40", "GOCTR_BENCH_PRED_BATCHES": "10"}, gpus=1)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][0])
    wp = d["without_preload"]
    assert d["n_gpus"] == 1 and d["preload"]["scratch_model_training_steps"] == 40
    assert wp and len(wp["timed_regions_ms"]) == 9 and wp["ms_per_step"] > 0
    assert wp["value"] == pytest.approx(8192 / (wp["ms_per_step"] * 1e-3), rel=0.02)


def test_bench_mlp100k_line_shape(stub):
    """BASELINE configs[0] has a workload of its own: 79 948 x 281 rows, batch 200, 20 epochs (the stub trains nothing)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(GOCTR_LIB=stub)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "mlp100k", "--no-cpu-baseline", "--regions", "2"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["updates"] == 20 * 400 and d["steps"] == 20 and d["dtype"] == "f64" and "configs[0]" in d["config"]["workload"]
    assert d["roofline"]["launch_floor"]["launches_per_update"] == 3 and d["reference_readme_s"] == 28.0
