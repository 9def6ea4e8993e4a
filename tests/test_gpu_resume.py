"""GPU: checkpoint / resume through the on-disk JSON.  The reference's dinModel / mlpModel JSON
(din.go:41-80, dnn.go:38-61) holds the weights only; Marshal(optimizer=True) adds the Adam moments and the step
counter under extra keys.  A model stopped after k steps, written out, read back and continued must land on the
SAME BITS as one that was never stopped (dropout on, so the dropout stream position has to survive too)."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(rng, rows, U, T, D, Cc, V):
    ub = rng.integers(-1, V, size=(rows, T)).astype(np.int32)
    items = rng.integers(0, V, size=rows).astype(np.int32)
    uf = rng.random((rows, U), dtype=np.float32)
    cf = rng.random((rows, Cc), dtype=np.float32)
    y = (rng.random(rows) < 0.4).astype(np.float32)
    emb = (rng.standard_normal((V, D)) * 0.25).astype(np.float32)
    return ub, items, uf, cf, y, emb


@pytest.mark.parametrize("kind", ["din", "youtube"])
def test_resume_is_bit_exact(kind):
    from goctr_amd import capi, model as gm
    rng = np.random.default_rng(11)
    U, T, D, Cc, V, B = 52, 20, 16, 53, 500, 256
    ub, items, uf, cf, y, emb = _data(rng, 6 * B, U, T, D, Cc, V)
    tab = gm.EmbeddingTable(emb)
    ds = gm.Dataset.ids(ub, items, uf, cf, y)
    cls, from_json = (gm.DinNet, gm.NewDinNetFromJson) if kind == "din" else (gm.YoutubeDnn, gm.NewYoutubeDnnFromJson)
    cfg = capi.default_train_cfg(batch=B, epochs=1)
    cfg.dropout_mode, cfg.p0, cfg.p1, cfg.seed = 2, 0.05, 0.05, 1234   # 2 = on-device hash stream
    names = cls(U, T, D, D, Cc).Learnable()

    def flat(m):
        return np.concatenate([m.get_weights(n).ravel() for n in names])

    a = cls(U, T, D, D, Cc).init_gaussian(np.random.default_rng(1))
    gm.train_steps(a, ds, cfg, 6, emb=tab)
    capi.sync()

    b = cls(U, T, D, D, Cc).init_gaussian(np.random.default_rng(1))
    gm.train_steps(b, ds, cfg, 3, emb=tab)
    capi.sync()
    blob = b.Marshal(optimizer=True)
    d = json.loads(blob)
    assert d["adamStep"] == 3 and len(d["mlp0_m"]) == len(d["mlp0"])
    assert any(v != 0.0 for v in d["mlp1_v"])
    b.close()
    c = from_json(blob)
    assert c.step == 3
    gm.train_steps(c, ds, cfg, 3, first_batch=3, emb=tab)
    capi.sync()
    assert np.array_equal(flat(a), flat(c))

    # and without the optimizer keys (the reference's own file) the continuation differs: Adam restarts
    e = from_json(json.dumps({k: v for k, v in d.items() if not (k.endswith("_m") or k.endswith("_v") or k == "adamStep")}))
    assert e.step == 0
    gm.train_steps(e, ds, cfg, 3, first_batch=3, emb=tab)
    capi.sync()
    assert not np.array_equal(flat(a), flat(e))


def test_moments_round_trip():
    from goctr_amd import model as gm
    m = gm.DinNet(7, 5, 4, 4, 3)
    rng = np.random.default_rng(0)
    for n in m.Learnable():
        for which in (0, 1):
            v = rng.standard_normal(m._shape(n)).astype(np.float32)
            m.set_moments(n, which, v)
            assert np.array_equal(m.get_moments(n, which), v)
    m.step = 41
    assert m.step == 41
    with pytest.raises(Exception):
        m.set_moments("mlp0", 2, np.zeros(m._shape("mlp0"), np.float32))
    with pytest.raises(Exception):
        m.set_moments("mlp0", 0, np.zeros(3, np.float32))
