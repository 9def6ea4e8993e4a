"""emb_io -- go-ctr's embedding text format (host-side, no device work):

    vector.Save      feature/embedding/model/modelutil/vector/vector.go:40-67    "word v0 v1 ... \n", values as %f
    emb.Load/parse   feature/embedding/emb/embedding.go:73-131                   one embedding per line, Fields-split;
                                                                                  lines that START with a space are skipped
The loaded (word, vector) pairs feed search.New (the device k-NN searcher) or model.EmbeddingTable.
"""
from __future__ import annotations

import math

import numpy as np


def Save(f, words, mat) -> None:
    """vector.Save: `fmt.Fprintf("%v ", word)`, then `"%f "` per element (6 decimals), then a newline"""
    mat = np.asarray(mat, np.float64)
    if len(words) != mat.shape[0]:
        raise ValueError(f"different for length of dic and row of matrix: {len(words)}, {mat.shape[0]}")
    out = []
    for w, row in zip(words, mat):
        out.append(f"{w} " + "".join(f"{v:f} " for v in row) + "\n")
    f.write("".join(out))


def parse_line(line: str):
    """emb.parseLine (embedding.go:108-131) -> (word, vector float64, norm)"""
    parts = line.split()
    if len(parts) < 2:
        raise ValueError("Must be over 2 lenghth for word and vector elems")
    vec = np.array([float(x) for x in parts[1:]], np.float64)
    n = 0.0
    for v in vec:                    # embutil.Norm (embutil.go:21-27): sequential sum, then sqrt
        n += v * v
    return parts[0], vec, math.sqrt(n)


def Load(text: str):
    """emb.Load (embedding.go:73-106): list of (word, vector); all vectors must share one dimension"""
    embs = []
    for line in text.splitlines():
        if line.startswith(" "):     # embedding.go:91-93
            continue
        w, v, _ = parse_line(line)
        if embs and v.size != embs[0][1].size:
            raise ValueError(f"dimension for all vectors must be the same: {embs[0][1].size} but got {v.size}")
        embs.append((w, v))
    return embs
