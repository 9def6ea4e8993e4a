"""Roofline provenance on CPU: scripts/prof_summarize.py keys rocprofv3 rows on the FULL kernel symbol (template arguments
included) and bench.py's pmc_entry() only returns the entry of exactly (phase, symbol, grid) -- round 2's summaries mapped two
kernels to one name and picked "the largest grid", and the driver line quoted the forward kernel's counters for the training one."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_kernel_symbols_keep_their_template_arguments():
    ps = load(os.path.join(ROOT, "scripts", "prof_summarize.py"), "prof_summarize")
    cases = {
        "void goctr::ctr_chain_x3_kernel<9, false>(goctr::ChainX3Args)": "ctr_chain_x3_kernel<9,false>",
        "void goctr::ctr_chain_x3_kernel<9, true>(goctr::ChainX3Args)": "ctr_chain_x3_kernel<9,true>",
        "void goctr::ctr_serve16_kernel<4, 2, 10>(goctr::AttnArgs, goctr::ChainArgs)": "ctr_serve16_kernel<4,2,10>",
        "void goctr::reduce_attn_kernel<4, 4, 2>(goctr::ReduceAdamArgs, goctr::AttnArgs, int)": "reduce_attn_kernel<4,4,2>",
        "void (anonymous namespace)::w2v_hogwild_kernel<16, 0, 0>((anonymous namespace)::W2vDev, int, long long const*, long long const*, long long const*, (anonymous namespace)::HogHot)": "w2v_hogwild_kernel<16,0,0>",
        "goctr::reduce_adam_kernel(goctr::ReduceAdamArgs)": "reduce_adam_kernel",
        "void goctr::gemm_nn_rows_kernel<float, goctr::EpiStore, 4>(float const*, int, float const*, int, int, int, int, int, int, goctr::EpiStore)": "gemm_nn_rows_kernel<float,EpiStore,4>",
    }
    for raw, want in cases.items():
        assert ps.symbol(raw) == want, (raw, ps.symbol(raw))
    assert ps.symbol("__amd_rocclr_fillBufferAligned") is None        # not this library's
    # the two chain instantiations are different keys
    assert ps.symbol("void goctr::ctr_chain_x3_kernel<9, false>(goctr::ChainX3Args)") != ps.symbol("void goctr::ctr_chain_x3_kernel<9, true>(goctr::ChainX3Args)")


def test_pmc_entry_returns_only_the_exact_phase_symbol_and_grid():
    sys.path.insert(0, ROOT)
    import bench
    import glob
    doc = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_din_kernels.json")))[-1]))      # (the one bench.py reads: the newest)
    train = doc["phases"]["train"]
    key = next(k for k in train if k.startswith("ctr_chain_x3_kernel<9,false>@"))
    sym, grid = key.split("@")
    e = bench.pmc_entry("din", "train", sym, int(grid))
    assert e and e.get("hbm_bytes") == train[key]["hbm_bytes"]
    # the forward-only instantiation is never substituted for the training one, nor another grid, nor another phase
    assert not bench.pmc_entry("din", "train", "ctr_chain_x3_kernel<9,true>", int(grid))
    assert not bench.pmc_entry("din", "train", sym, int(grid) * 2)
    assert not bench.pmc_entry("din", "predict", sym, int(grid))
    assert not bench.pmc_entry("din", "train", "no_such_kernel", None)


def test_step_total_is_the_sum_of_the_replayed_steps_launches():
    """the whole step's memory-side traffic is recomputable from ONE tracked file -- every launch of the
    graph-replayed step (chain, weight gradients, reduce_attn: the last one has counters since the PMC passes run the pipelined
    kernels eagerly) carries hbm_bytes, and step_total is their sum over the SURVEY 8(d) algorithmic bytes"""
    import glob
    for wl, per_sample, batch in (("din", 3900, 8192), ("youtube", 13680, 16384)):
        doc = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{wl}_kernels.json")))[-1]))
        st = doc["step_total"]
        ents = [doc["phases"]["train"][k] for k in st["kernels"]]
        assert len(ents) == 3 and all(e.get("hbm_bytes") for e in ents) and any(e["kernel"].startswith("reduce_attn_kernel") for e in ents)
        assert st["sum_hbm_bytes"] == sum(e["hbm_bytes"] for e in ents)
        assert abs(st["sum_avg_us"] - sum(e["avg_us"] for e in ents)) < 1e-6
        assert st["algorithmic_bytes"] == per_sample * batch
        assert abs(st["traffic_ratio"] - st["sum_hbm_bytes"] / st["algorithmic_bytes"]) < 1e-3
        # the predict phase has SQ counters and an L2 hit rate too
        pred = [e for e in doc["phases"]["predict"].values() if e["kernel"].startswith(("ctr_chain_x3_kernel", "ctr_fwd4_kernel")) and e.get("calls", 0) > 10]
        assert pred and pred[0].get("sq") and pred[0].get("l2_hit_rate") is not None
