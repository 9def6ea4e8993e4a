"""search -- host mirror of go-ctr's embedding k-NN searcher (feature/embedding/search/search.go) over the HIP engine.

    Neighbor / Neighbors        search.go:27-33
    New(*embs) / Searcher       search.go:52-63
    Searcher.SearchInternal     search.go:65-83
    Searcher.SearchVector       search.go:85-90
    Searcher.Search             search.go:92-134

All arithmetic (norms, cosine scores, top-k selection) runs in libgoctr_hip.so (csrc/search.hip); this module only
maps words <-> item indices.  Batched entry points (search_vectors) score many queries per call.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi


@dataclass
class Neighbor:
    """search.go:27-31"""
    Word: str = ""
    Rank: int = 0
    Similarity: float = 0.0


class Searcher:
    def __init__(self, words, vectors):
        capi.init()
        self.words = list(words)
        vec = np.ascontiguousarray(vectors, np.float64)
        if vec.ndim != 2 or vec.shape[0] != len(self.words) or vec.shape[0] == 0:
            raise ValueError("embeddings must be a non-empty [V, D] matrix with one word per row")   # emb.Validate
        self.dim = vec.shape[1]
        self._index = {}
        for i, w in enumerate(self.words):          # SearchInternal takes the FIRST item with that word (:67-72)
            self._index.setdefault(w, i)
        self._vec = vec
        self._h = C.c_void_p()
        capi.check(capi.load().goctr_searcher_create(capi.ptr(vec, C.c_double), C.c_int64(vec.shape[0]), C.c_int(self.dim),
                                                     C.byref(self._h)))

    # ---- batched core
    def search_vectors(self, queries, k, ignore=None):
        """Q queries at once -> (idx [Q,k] int64, sim [Q,k] float64, count [Q]); idx -1 = empty neighbour."""
        q = np.ascontiguousarray(queries, np.float64).reshape(-1, self.dim)
        Q = q.shape[0]
        idx = np.empty((Q, k), np.int64)
        sim = np.empty((Q, k), np.float64)
        cnt = np.empty(Q, np.int32)
        ig = None if ignore is None else np.ascontiguousarray(ignore, np.int64)
        capi.check(capi.load().goctr_searcher_search(self._h, capi.ptr(q, C.c_double), C.c_int(Q), C.c_int(k),
                                                     capi.ptr(ig, C.c_int64), capi.ptr(idx, C.c_int64),
                                                     capi.ptr(sim, C.c_double), capi.ptr(cnt, C.c_int32)))
        return idx, sim, cnt

    def _neighbors(self, idx, sim, cnt):
        out = []
        for r in range(int(cnt)):
            i = int(idx[r])
            out.append(Neighbor(self.words[i], r + 1, float(sim[r])) if i >= 0 else Neighbor())
        return out

    # ---- the reference's methods
    def Search(self, query_vector, k, *ignoreWord):
        ig = -1
        if ignoreWord:
            if len(ignoreWord) > 1:
                raise ValueError("the device path skips one item per query")
            ig = self._index.get(ignoreWord[0], -1)
        idx, sim, cnt = self.search_vectors(query_vector, k, None if ig < 0 else [ig])
        return self._neighbors(idx[0], sim[0], cnt[0])

    def SearchVector(self, query, k):
        return self.Search(query, k)

    def SearchInternal(self, word, k):
        if word not in self._index:
            raise KeyError(f"{word} is not found in searcher")                     # search.go:73-75
        return self.Search(self._vec[self._index[word]], k, word)

    def close(self):
        if getattr(self, "_h", None):
            capi.load().goctr_searcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def New(*embs):
    """search.New(embs...): embs = (word, vector) pairs"""
    return Searcher([w for w, _ in embs], [v for _, v in embs])
