"""host-side embedding text format, pinned on the reference's embedding_test.go"""
import io
import json
import math
import os

import numpy as np
import pytest

from goctr_amd import emb_io

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kats.json")))["emb_load"]


def test_reference_kats():
    assert len(emb_io.Load(KATS["contents_load"])) == KATS["item_size"]
    assert len(emb_io.Load(KATS["contents_parse"])) == KATS["num_vector"]
    w, v, n = emb_io.parse_line(KATS["line"])
    e = KATS["expected"]
    assert w == e["Word"] and v.size == e["Dim"] and v.tolist() == e["Vector"] and n == math.sqrt(5.0)


def test_save_load_round_trip_and_quirks():
    words = ["a", "b", "c"]
    mat = np.array([[0.5, -1.25, 3.0], [1e-7, 2.0, -0.0], [123456.789, 0.1, 1.0 / 3.0]])
    f = io.StringIO()
    emb_io.Save(f, words, mat)
    text = f.getvalue()
    assert text.splitlines()[0] == "a 0.500000 -1.250000 3.000000 "            # %f and the trailing blank (vector.go:55-57)
    back = emb_io.Load(text)
    assert [w for w, _ in back] == words
    assert np.allclose(np.stack([v for _, v in back]), mat, atol=5e-7)        # %f keeps 6 decimals
    assert emb_io.Load(" skipped 1 2 3\nkept 1 2 3")[0][0] == "kept"           # lines starting with a space are skipped
    with pytest.raises(ValueError):
        emb_io.Load("x 1 2 3\ny 1 2")                                          # Validate: one dimension for all
    with pytest.raises(ValueError):
        emb_io.parse_line("lonely")
    with pytest.raises(ValueError):
        emb_io.Save(io.StringIO(), ["a"], mat)
