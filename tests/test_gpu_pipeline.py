"""The path bench.py TIMES, against the oracle and against its own unpipelined / ungraphed variants.

bench.py's training region is goctr_train_steps over an id-mode dataset with the reference's dropout: steps replayed as
16- / 4- / 2-step hipGraphs whose steps are PIPELINED (reduce_attn_kernel computes step n+1's gather and gates inside step
n's last launch and picks up the updated att0 through an intra-launch flag, csrc/ctr_kernels.h).  The one-step full-size
tests (test_gpu_fullsize.py) never execute that hand-off; these do, at the sizes of BASELINE configs[2] / configs[3]:

  * >= 20 consecutive steps through goctr_train_steps vs the oracle stepping the same batches (model/model.go:96-211:
    forward, BCE, backward, gorgonia Adam per batch) -- per-step cost, final weights, att0;
  * GOCTR_PIPELINE=0/1 and GOCTR_GRAPH_STEPS=0/1 bit-equality IN ID MODE at B = 8192 (the dense-mode test of
    test_gpu_ctr.py never pipelines);
  * DIN predict launches of >= 16 384 rows (ctr_chain_x3_kernel<9,true>, the kernel behind recommend_qps) vs the oracle.
"""
import os

import numpy as np
import pytest

from test_gpu_fullsize import LOGIT_TOL, LOSS_TOL, models, synth

pytestmark = pytest.mark.gpu


def _free_running_steps_vs_oracle(oracle, kind, U, T, D, Cc, V, B, steps, seed, pdrop):
    from goctr_amd import capi, model as gm
    rng = np.random.default_rng(seed)
    rows = steps * B
    emb, ub, it, uf, cf, Y = synth(rng, rows, U, T, D, Cc, V)
    om, dm = models(oracle, kind, U, T, D, Cc, rng, 0.15)
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, Y)
    cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2, p0=pdrop, p1=pdrop, seed=77)
    costs = gm.train_steps(dm, ds, cfg, steps, emb=tab, want_costs=True)     # e.g. 20 = one 16-step + one 4-step graph

    # float32 noise floor of each gradient tensor: the oracle's own distance from a float64 evaluation (batch 0, no dropout)
    X0 = oracle.assemble_rows(emb, ub[:B], it[:B], uf[:B], cf[:B])
    _, rg0, _ = om.loss_grad(X0, Y[:B], B=B)
    _, g64, _ = om.loss_grad_f64(X0, Y[:B], B=B)
    names = ["W0", "W1", "W2"] + (["att0"] if kind == 0 else [])
    noise = {n: float(np.max(np.abs(np.asarray(rg0[n], np.float64) - g64[n]))) for n in names}
    near_zero = {n: np.zeros(getattr(om, n).shape, bool) for n in names}
    l2 = float(cfg.l2)

    st, ref = None, []
    for k in range(steps):
        lo = k * B
        X = oracle.assemble_rows(emb, ub[lo:lo + B], it[lo:lo + B], uf[lo:lo + B], cf[lo:lo + B])
        c, g, _ = om.loss_grad(X, Y[lo:lo + B], B=B, drop=dict(mode=2, p0=pdrop, p1=pdrop, seed=77, step=k))
        for n in names:
            # the quantity whose SIGN Adam's normalised update follows (gorgonia: L2 first): an entry inside the float32 noise
            # of zero gets an ill-conditioned update (+-lr on the first step), on the device and in the oracle alike
            near_zero[n] |= np.abs(g[n].reshape(near_zero[n].shape) + l2 * getattr(om, n)) <= 8 * noise[n]
        st = om.adam_step(g, state=st, batch=B)
        ref.append(c)
    ref = np.array(ref, np.float32)
    dcost = np.abs(costs - ref)
    print(f"kind {kind}: per-step |cost - oracle| max {dcost.max():.2e} (first {dcost[0]:.2e}, last {dcost[-1]:.2e})")
    assert dcost[0] <= LOSS_TOL
    assert np.max(dcost) <= LOSS_TOL, dcost.tolist()

    pairs = [("mlp0", "W0"), ("mlp1", "W1"), ("mlp2", "W2")] + ([("att0", "att0")] if kind == 0 else [])
    for dn, on in pairs:
        d = np.abs(dm.get_weights(dn).reshape(getattr(om, on).shape) - getattr(om, on))
        out = d > 1e-5
        frac, unexplained = float(out.mean()), int(np.sum(out & ~near_zero[on]))
        print(f"  {on}: max {d.max():.2e}  q99.9 {np.quantile(d, 0.999):.2e}  entries > 1e-5: {int(out.sum())} ({frac:.2%}), "
              f"of them NOT near a zero gradient: {unexplained};  near-zero entries {float(near_zero[on].mean()):.2%}")
        # every entry the two runs disagree on passed, at some step, within float32 noise of a zero gradient ...
        assert unexplained == 0, (on, unexplained)
        # ... and there are few of them
        assert np.quantile(d, 0.999) <= 1e-5 or frac <= 0.01, (on, frac)
    if kind == 0:
        assert np.max(np.abs(dm.get_weights("att0").ravel() - om.att0.ravel())) <= 1e-6


def test_cfg3_din_20_pipelined_graph_steps_vs_oracle(oracle):
    """BASELINE configs[2] at its stated size, exactly the launch sequence bench.py times (K = 20: a 16-step and a 4-step graph)"""
    _free_running_steps_vs_oracle(oracle, 0, 52, 50, 16, 53, 26744, 8192, 20, 300, 0.005)


def test_cfg4_youtube_6_pipelined_graph_steps_vs_oracle(oracle):
    """BASELINE configs[3] per-GPU slice at its stated size (2.56 GB table): a 4-step and a 2-step graph"""
    _free_running_steps_vs_oracle(oracle, 1, 52, 50, 64, 53, 10_000_000, 16384, 6, 301, 0.003)


@pytest.mark.parametrize("knob", ["GOCTR_PIPELINE", "GOCTR_GRAPH_STEPS", "GOCTR_NO_GRAPH", "GOCTR_CHAIN_ATTN_BWD", "GOCTR_CHAIN_TILE_SUMS",
                                  "GOCTR_XCD_AFFINE", "GOCTR_TN_WT", "GOCTR_GATE_FAC", "GOCTR_ATT0_EARLY"])
def test_id_mode_pipelined_graphs_equal_unpipelined_and_eager(knob):
    """id mode, B = 8192 (377 reduce blocks beside 2048 attention workgroups in the merged launch), dropout on, 39 steps =
    16 + 16 + 4 + 2 + 1: the default path and the path with the knob flipped must land on the same bits"""
    from goctr_amd import capi, model as gm
    U, T, D, Cc, V, B, steps = 52, 50, 16, 53, 26744, 8192, 39
    rng = np.random.default_rng(310)
    emb, ub, it, uf, cf, Y = synth(rng, 8 * B, U, T, D, Cc, V)
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, Y)
    res = []
    # (GOCTR_CHAIN_ATTN_BWD=0: the att0 gradient's per-sample terms from the separate attn_bwd_kernel instead of the chain
    # kernel's tail -- same arithmetic in the same order)
    # (round 6 -- GOCTR_XCD_AFFINE=0: workgroup b takes tile / sample group b instead of its XCD's; GOCTR_TN_WT=0: plain slab stores:
    # same bits.  GOCTR_CHAIN_TILE_SUMS=0: dW2 and the att0 terms leave the chain launch as operands and are summed by the
    # weight-gradient launch's MFMA problems instead of per tile in the chain launch -- another float32 summation order, compared
    # at 2e-6 below; the GOCTR_CHAIN_ATTN_BWD comparison is between the two stored-terms paths, i.e. with the tile sums off.
    # GOCTR_GATE_FAC=0: the attention forward stores gate and similarity weight and the chain launch's tail forms (g (1 - g)) w itself
    # instead of reading that factor ready-made -- the same statement on the same values, the same bits.
    # GOCTR_ATT0_EARLY=0: att0's tile sums go through slabs and the last launch's reduce block, whose flag the attention wavefronts
    # wait for, instead of one workgroup of the weight-gradient launch adding them up and applying Adam -- the same additions in the
    # same order, the same bits)
    flipped = {"GOCTR_PIPELINE": "0", "GOCTR_GRAPH_STEPS": "0", "GOCTR_NO_GRAPH": "1", "GOCTR_CHAIN_ATTN_BWD": "0",
               "GOCTR_CHAIN_TILE_SUMS": "0", "GOCTR_XCD_AFFINE": "0", "GOCTR_TN_WT": "0", "GOCTR_GATE_FAC": "0", "GOCTR_ATT0_EARLY": "0"}[knob]
    for val in (None, flipped):
        if val is not None:
            os.environ[knob] = val
        if knob == "GOCTR_CHAIN_ATTN_BWD":
            os.environ["GOCTR_CHAIN_TILE_SUMS"] = "0"
        try:
            m = gm.DinNet(U, T, D, D, Cc)
            r = np.random.default_rng(311)
            m.set_weights("mlp0", (r.standard_normal((U + 2 * D + Cc, 200)) * 0.15).astype(np.float32))
            m.set_weights("mlp1", (r.standard_normal((200, 80)) * 0.15).astype(np.float32))
            m.set_weights("mlp2", (r.standard_normal((80, 1)) * 0.15).astype(np.float32))
            m.set_weights("att0", (1 + 0.3 * r.standard_normal(T)).astype(np.float32))
            cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2, p0=0.005, p1=0.005, seed=5)
            costs = gm.train_steps(m, ds, cfg, steps, emb=tab, want_costs=True)
            res.append((costs, m.get_weights("mlp0"), m.get_weights("mlp1"), m.get_weights("att0")))
        finally:
            os.environ.pop(knob, None)
            if knob == "GOCTR_CHAIN_ATTN_BWD":
                os.environ.pop("GOCTR_CHAIN_TILE_SUMS", None)
    assert np.all(np.isfinite(res[0][0]))
    for a, b in zip(res[0], res[1]):
        if knob == "GOCTR_CHAIN_TILE_SUMS":
            # 39 Adam steps of lr 0.01 on gradients that differ in rounding: an entry whose gradient passes within float32 noise of
            # zero takes a different +-lr step (the same effect as against the oracle, tests/test_gpu_fullsize.py)
            assert np.mean(np.abs(a - b) <= 1e-5) >= 0.999 and np.max(np.abs(a - b)) <= 0.05
        else:
            assert np.array_equal(a, b)


@pytest.mark.parametrize("att", [0, 1])
def test_din_predict_of_16384_row_launches_vs_oracle(oracle, att):
    """recommend_qps's kernel: predict launches that give every CU a 32-row tile run ctr_chain_x3_kernel<9,true>, one
    persistent workgroup per CU over the launch's row tiles.  8 batches of 4096 form one 32 768-row launch (4 tiles per
    workgroup), the next 4 a 16 384-row one, then a 2-batch launch (8192 rows: one tile each); the last, short batch runs
    ctr_fwd16_kernel"""
    from goctr_amd import model as gm
    U, T, D, Cc, V = 52, 50, 16, 53, 26744
    rows = 4096 * 14 + 777
    rng = np.random.default_rng(320 + att)
    emb, ub, it, uf, cf, _ = synth(rng, rows, U, T, D, Cc, V)
    om = oracle.CtrModel(0, U, T, D, Cc, att=att)
    om.W0[:] = (rng.standard_normal(om.W0.shape) * 0.15).astype(np.float32)
    om.W1[:] = (rng.standard_normal(om.W1.shape) * 0.15).astype(np.float32)
    om.W2[:] = (rng.standard_normal(om.W2.shape) * 0.15).astype(np.float32)
    om.att0[:] = (1 + 0.3 * rng.standard_normal(T)).astype(np.float32)
    dm = gm.DinNet(U, T, D, D, Cc, att=att)
    for n, w in (("mlp0", om.W0), ("mlp1", om.W1), ("mlp2", om.W2), ("att0", om.att0)):
        dm.set_weights(n, w)
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, None)
    y = gm.predict_dataset(dm, ds, 4096, emb=tab)
    X = oracle.assemble_rows(emb, ub, it, uf, cf)
    ry = om.predict(X, 4096)
    assert y.shape == (rows,)
    assert np.max(np.abs(y[:32768] - ry[:32768])) <= LOGIT_TOL          # the 32 768-row launch
    assert np.max(np.abs(y - ry)) <= LOGIT_TOL


@pytest.mark.parametrize("kind,D,PB,rows", [("din", 16, 1100, 8 * 1100), ("din", 16, 4096, 8 * 4096), ("din", 16, 1037, 8 * 1037 - 5),
                                            ("din", 16, 4096, 3 * 8 * 4096 + 100), ("youtube", 64, 2048, 8 * 2048), ("youtube", 64, 1100, 3 * 8 * 1100 - 9)])
def test_four_wavefront_forward_kernel_against_the_eight_wavefront_one(kind, D, PB, rows):
    """forward-only launches at Ip <= 144: four wavefronts per 32-row tile and two workgroups per CU (ctr_fwd4.h, default)
    against the 8-wavefront kernel (GOCTR_FWD4=0).  The two add the layer-1 partial sums in different orders (four partials of
    four k chunks / seven of two), so the scores agree to float32 rounding, not bit for bit; each is inside the 1e-5 bar against
    the oracle (test_gpu_fullsize.py runs the default).  Both walk their tiles as persistent workgroups: 275 tiles (a second trip
    for 19 of the 8-wavefront kernel's 256 workgroups, none for the 512 four-wavefront ones), 1024 tiles (4 / 2 trips each), a
    launch whose last tile is partly past the dataset's end, and a dataset of several launches; YouTube-DNN at Ip = 240 (the
    exchange one H2 tile at a time through two 16 KiB halves)"""
    from goctr_amd import model as gm
    U, T, Cc, V = 52, 50, 53, 5000
    rng = np.random.default_rng(rows + 2)
    emb, ub, it, uf, cf, _ = synth(rng, rows, U, T, D, Cc, V)
    dm = gm.DinNet(U, T, D, D, Cc) if kind == "din" else gm.YoutubeDnn(U, T, D, D, Cc)
    r = np.random.default_rng(7)
    dm.set_weights("mlp0", (r.standard_normal((U + 2 * D + Cc, 200)) * 0.15).astype(np.float32))
    dm.set_weights("mlp1", (r.standard_normal((200, 80)) * 0.15).astype(np.float32))
    dm.set_weights("mlp2", (r.standard_normal((80, 1)) * 0.15).astype(np.float32))
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, it, uf, cf, None)
    ys = []
    for knob in (None, "0"):
        if knob is not None:
            os.environ["GOCTR_FWD4"] = knob
        try:
            ys.append(gm.predict_dataset(dm, ds, PB, emb=tab))
        finally:
            os.environ.pop("GOCTR_FWD4", None)
    assert ys[0].shape == (rows,) and np.all((ys[0] > 0) & (ys[0] < 1))
    assert np.max(np.abs(ys[0] - ys[1])) <= 1e-6
    assert not np.array_equal(ys[0], np.full(rows, ys[0][0]))


@pytest.mark.parametrize("between", ["nothing", "set_att0", "set_rows", "predict", "jump", "other_steps", "train_emb"])
def test_h0_carried_from_the_previous_call_only_when_nothing_touched_it(between):
    """the last launch of a goctr_train_steps call has computed the next batch's h0 / gates; the next call skips its first
    attn_fwd when it starts at exactly that batch and no entry point that could touch weights, table, dataset or workspace ran
    in between (goctr_model::H0Carry).  10 + 10 steps with something in between, carry on (default) against GOCTR_H0_CARRY=0:
    the same bits -- for `nothing` the carry is used, for every other case it must have been dropped"""
    from goctr_amd import capi, model as gm
    import ctypes as C
    U, T, D, Cc, V, B = 52, 50, 16, 53, 5000, 512
    rng = np.random.default_rng(77)
    emb, ub, it, uf, cf, Y = synth(rng, 24 * B, U, T, D, Cc, V)
    res = []
    for knob in (None, "0"):
        if knob is not None:
            os.environ["GOCTR_H0_CARRY"] = knob
        try:
            tab = gm.EmbeddingTable(emb)
            ds = gm.Dataset.ids(ub, it, uf, cf, Y)
            m = gm.DinNet(U, T, D, D, Cc)
            r = np.random.default_rng(78)
            m.set_weights("mlp0", (r.standard_normal((U + 2 * D + Cc, 200)) * 0.15).astype(np.float32))
            m.set_weights("mlp1", (r.standard_normal((200, 80)) * 0.15).astype(np.float32))
            m.set_weights("mlp2", (r.standard_normal((80, 1)) * 0.15).astype(np.float32))
            m.set_weights("att0", (1 + 0.3 * r.standard_normal(T)).astype(np.float32))
            cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2, p0=0.01, p1=0.01, seed=3)
            c1 = gm.train_steps(m, ds, cfg, 10, first_batch=0, emb=tab, want_costs=True)
            nxt = 10
            if between == "set_att0":
                m.set_weights("att0", (1 + 0.3 * np.random.default_rng(5).standard_normal(T)).astype(np.float32))
            elif between == "set_rows":
                new = (np.random.default_rng(6).standard_normal((64, D)) * 0.3).astype(np.float32)
                capi.check(capi.load().goctr_emb_set_rows(tab._h, C.c_int64(0), C.c_int64(64), capi.ptr(new, C.c_float)))
            elif between == "predict":
                gm.predict_dataset(m, ds, 512, emb=tab)
            elif between == "jump":
                nxt = 3
            elif between == "other_steps":
                gm.train_steps(m, ds, cfg, 2, first_batch=20, emb=tab)
            elif between == "train_emb":
                # (the first call's attention launches left the one factor (g (1 - g)) w, AttnArgs::fac; with the table trainable the
                # backward reads gate and weight themselves: what the first call's last launch computed must not be carried)
                m.set_embedding_training(0.05)
            c2 = gm.train_steps(m, ds, cfg, 10, first_batch=nxt, emb=tab, want_costs=True)
            res.append((c1, c2, m.get_weights("mlp0"), m.get_weights("att0")))
        finally:
            os.environ.pop("GOCTR_H0_CARRY", None)
    assert np.all(np.isfinite(res[0][1]))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("change_betas", [False, True])
def test_carried_start_without_costs_needs_no_state_preparation(change_betas):
    """a carried start whose caller reads no costs also skips the state-preparation launch (the previous call's last loss
    block left cursor and Adam bias corrections): three calls of 7 steps each, continuing, against GOCTR_H0_CARRY=0 -- and with
    other Adam betas in the second call, which must bring the preparation back (the corrections depend on them)"""
    from goctr_amd import capi, model as gm
    U, T, D, Cc, V, B = 52, 50, 16, 53, 5000, 512
    rng = np.random.default_rng(81)
    emb, ub, it, uf, cf, Y = synth(rng, 24 * B, U, T, D, Cc, V)
    res = []
    for knob in (None, "0"):
        if knob is not None:
            os.environ["GOCTR_H0_CARRY"] = knob
        try:
            tab = gm.EmbeddingTable(emb)
            ds = gm.Dataset.ids(ub, it, uf, cf, Y)
            m = gm.DinNet(U, T, D, D, Cc)
            r = np.random.default_rng(82)
            m.set_weights("mlp0", (r.standard_normal((U + 2 * D + Cc, 200)) * 0.15).astype(np.float32))
            m.set_weights("mlp1", (r.standard_normal((200, 80)) * 0.15).astype(np.float32))
            m.set_weights("mlp2", (r.standard_normal((80, 1)) * 0.15).astype(np.float32))
            m.set_weights("att0", (1 + 0.3 * r.standard_normal(T)).astype(np.float32))
            cfg = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2, p0=0.01, p1=0.01, seed=3)
            gm.train_steps(m, ds, cfg, 7, first_batch=0, emb=tab)
            cfg2 = capi.default_train_cfg(batch=B, epochs=1, dropout_mode=2, p0=0.01, p1=0.01, seed=3)
            if change_betas:
                cfg2.beta1, cfg2.beta2 = 0.8, 0.99
            gm.train_steps(m, ds, cfg2, 7, first_batch=7, emb=tab)
            last = gm.train_steps(m, ds, cfg2, 7, first_batch=14, emb=tab, want_costs=True)
            res.append((last, m.get_weights("mlp0"), m.get_weights("mlp1"), m.get_weights("att0")))
        finally:
            os.environ.pop("GOCTR_H0_CARRY", None)
    assert np.all(np.isfinite(res[0][0]))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
