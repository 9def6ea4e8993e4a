"""Host mirror of go-ctr's item2vec entry point.

Reference: feature/embedding/wordemb.go:9-32 (TrainEmbedding), feature/embedding/model/model.go:24-31
(Model interface), model/word2vec/word2vec.go (Train :90, GenEmbeddingMap32 :298, WordVector :249),
corpus/memory/memory.go:53-102 (Load / IndexedDoc), corpus/dictionary/dictionary.go:70-81 (Add),
corpus/cpsutil/cpsutil.go:58-78 (MinCount / MaxCount filters), modelutil/subsample/subsample.go:24-52.
The dictionary (string -> id, counts), the init RNG and the sub-sampling trials are host-side here exactly
as they are host-side Go in the reference; the SkipGram / hierarchical-softmax updates run on the device
through include/goctr.h (goctr_w2v_*).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class Dictionary:
    """corpus/dictionary/dictionary.go:20-81"""

    def __init__(self):
        self.word2id, self.id2word, self.cfs = {}, [], []

    def Add(self, word):
        i = self.word2id.get(word)
        if i is None:
            self.word2id[word] = len(self.id2word)
            self.id2word.append(word)
            self.cfs.append(1)
        else:
            self.cfs[i] += 1

    def Len(self):
        return len(self.id2word)


class Word2Vec:
    """embedding/model.Model (model/model.go:24-31) backed by the device engine."""

    def __init__(self, window=5, dim=16, iter=1, optimizer="hs", min_count=5, max_count=-1, init_lr=0.025,
                 subsample_threshold=1e-3, deterministic=False, streams=8192, slices=16, update_lr_batch=100000, rng=None,
                 model="skipgram", devices=0, exchange_every=0):
        # options.go:38-58 defaults; wordemb.go:10-18 fixes SkipGram + HS + DocInMemory
        self.window, self.dim, self.iter, self.optimizer = window, dim, iter, optimizer
        self.min_count, self.max_count, self.init_lr = min_count, max_count, init_lr
        self.threshold = subsample_threshold
        self.deterministic, self.streams = deterministic, streams
        self.slices = slices          # the reference's goroutine count (window-clipping units); workers = streams
        self.update_lr_batch = update_lr_batch
        self.model = model                                                  # options.go ModelType: skipgram | cbow
        self.devices = devices        # n of capi.init_devices: every pass runs data-parallel inside ONE call (goctr_w2v_cfg.devices)
        # data-parallel passes: words per rank between two all-reduces of the parameter deltas (0 = update_lr_batch, < 0 = once per
        # pass; goctr_w2v_cfg.exchange_every)
        self.exchange_every = exchange_every
        self.rng = rng or np.random.default_rng()
        self.dic = Dictionary()
        self.idoc = []
        self._h = None
        self.currentlr = init_lr

    # ---- corpus (memory.go:76-102 Load, :53-62 IndexedDoc)
    def load(self, words):
        for w in words:
            self.dic.Add(w)
            self.idoc.append(self.dic.word2id[w])
        return self

    def indexed_doc(self):
        cfs = np.asarray(self.dic.cfs, np.int64)
        idoc = np.asarray(self.idoc, np.int32)
        drop = cfs[idoc] < self.min_count                                   # cpsutil.go:72-76 (0 <= v && freq < v)
        if self.max_count > 0:
            drop |= cfs[idoc] > self.max_count                              # cpsutil.go:64-68
        return idoc[~drop]

    def _cfg(self):
        c = capi.W2vCfg()
        capi.load().goctr_w2v_cfg_default(C.byref(c))
        c.dim, c.window, c.optimizer = self.dim, self.window, 0 if self.optimizer == "hs" else 1
        c.model = 0 if self.model == "skipgram" else 1
        c.init_lr, c.min_lr = self.init_lr, self.init_lr * 1.0e-4           # options.go:42,49
        c.update_lr_batch = self.update_lr_batch
        c.deterministic, c.streams, c.slices = int(self.deterministic), self.streams, self.slices
        c.devices = self.devices
        c.exchange_every = self.exchange_every
        return c

    def create(self, counts, param0=None, aux0=None):
        capi.init()
        self.close()
        counts = np.ascontiguousarray(counts, np.int64)
        self.V = counts.size
        self._h = C.c_void_p()
        cfg = self._cfg()
        capi.check(capi.load().goctr_w2v_create(C.byref(cfg), C.c_int64(self.V), capi.ptr(counts, C.c_int64),
                                                C.byref(self._h)))
        if param0 is None:                                                  # word2vec.go:103-111
            param0 = (self.rng.random((self.V, self.dim)) - 0.5) / self.dim
        self.set_param(param0)
        if aux0 is None and self.optimizer != "hs":                         # optimizer.go:38-48
            aux0 = (self.rng.random((self.V, self.dim)) - 0.5) / self.dim
        if aux0 is not None:
            self.set_aux(aux0)
        return self

    def set_param(self, p):
        p = np.ascontiguousarray(p, np.float64)
        capi.check(capi.load().goctr_w2v_set_param(self._h, capi.ptr(p, C.c_double)))

    def set_aux(self, a):
        a = np.ascontiguousarray(a, np.float64)
        capi.check(capi.load().goctr_w2v_set_aux(self._h, capi.ptr(a, C.c_double)))

    def get_param(self):
        p = np.empty((self.V, self.dim), np.float64)
        capi.check(capi.load().goctr_w2v_get_param(self._h, capi.ptr(p, C.c_double)))
        return p

    def get_aux(self):
        rows = max(self.V - 1, 1) if self.optimizer == "hs" else self.V
        a = np.empty((rows, self.dim), np.float64)
        capi.check(capi.load().goctr_w2v_get_aux(self._h, capi.ptr(a, C.c_double)))
        return a

    def get_paths(self):
        total = C.c_int64(0)
        off = np.zeros(self.V + 1, np.int64)
        L = capi.load()
        capi.check(L.goctr_w2v_get_paths(self._h, capi.ptr(off, C.c_int64), None, None, C.c_int64(0), C.byref(total)))
        nodes = np.zeros(max(total.value, 1), np.int32)
        codes = np.zeros(max(total.value, 1), np.uint8)
        capi.check(L.goctr_w2v_get_paths(self._h, capi.ptr(off, C.c_int64), capi.ptr(nodes, C.c_int32),
                                         capi.ptr(codes, C.c_uint8), C.c_int64(total.value), C.byref(total)))
        return off, nodes[:total.value], codes[:total.value]

    def train_pass(self, doc, corpus_len, keep_mask=None, lr=None):
        doc = np.ascontiguousarray(doc, np.int32)
        km = np.ascontiguousarray(keep_mask, np.uint8) if keep_mask is not None else None
        lr_c = C.c_double(self.currentlr if lr is None else lr)
        capi.check(capi.load().goctr_w2v_train(self._h, capi.ptr(doc, C.c_int32), C.c_int64(doc.size),
                                               C.c_int64(corpus_len), capi.ptr(km, C.c_uint8), C.byref(lr_c)))
        self.currentlr = lr_c.value
        return lr_c.value

    def upload_doc(self, doc, keep_mask=None):
        doc = np.ascontiguousarray(doc, np.int32)
        km = np.ascontiguousarray(keep_mask, np.uint8) if keep_mask is not None else None
        capi.check(capi.load().goctr_w2v_upload_doc(self._h, capi.ptr(doc, C.c_int32), C.c_int64(doc.size),
                                                    capi.ptr(km, C.c_uint8)))

    def train_resident(self, corpus_len, lr=None):
        lr_c = C.c_double(self.currentlr if lr is None else lr)
        capi.check(capi.load().goctr_w2v_train_resident(self._h, C.c_int64(corpus_len), C.byref(lr_c)))
        self.currentlr = lr_c.value
        return lr_c.value

    # ---- Model interface
    def Train(self, words):
        """word2vec.go:90-149 + train :151-175"""
        self.load(words)
        cfs = np.asarray(self.dic.cfs, np.int64)
        self.create(cfs)
        doc = self.indexed_doc()
        keep_p = np.maximum(0.0, 1.0 - np.sqrt(self.threshold / cfs))       # subsample.go:28-43 (raw counts, Q14)
        for _ in range(self.iter):
            trial = self.rng.random(doc.size)
            keep = (keep_p[doc] > trial).astype(np.uint8)                   # Trial: subsample.go:45-52
            self.train_pass(doc, len(self.idoc), keep)
        return self

    def TrainIds(self, batches, capacity_words, seed=0):
        """word2vec.Train (word2vec.go:90-175) for integer tokens with everything resident: the token batches go to
        HBM as they arrive, dictionary / IndexedDoc / subsampling masks are built there (csrc/corpus.hip) and only the
        counts come back (for the host-side Huffman build)."""
        from .corpus import Corpus
        cps = Corpus(capacity_words, self.min_count, self.max_count).Load(batches)
        self.corpus = cps
        capi.init()
        self.close()
        self.V = cps.V
        self._h = C.c_void_p()
        cfg = self._cfg()
        capi.check(capi.load().goctr_w2v_create_from_corpus(C.byref(cfg), cps._h, C.byref(self._h)))
        self.set_param((self.rng.random((self.V, self.dim)) - 0.5) / self.dim)          # word2vec.go:103-111
        if self.optimizer != "hs":
            self.set_aux((self.rng.random((self.V, self.dim)) - 0.5) / self.dim)        # optimizer.go:38-48
        for it in range(self.iter):
            capi.check(capi.load().goctr_w2v_use_corpus(self._h, cps._h, C.c_double(self.threshold),
                                                        C.c_uint64(seed + it)))
            self.train_resident(cps.Len())
        return self

    def keep_mask(self, n):
        km = np.empty(n, np.uint8)
        capi.check(capi.load().goctr_w2v_get_keep_mask(self._h, capi.ptr(km, C.c_uint8), C.c_int64(n)))
        return km

    def WordVector(self):
        return self.get_param()                                             # HS: param rows (word2vec.go:249-271)

    def export_f32(self):
        out = np.empty((self.V, self.dim), np.float32)
        capi.check(capi.load().goctr_w2v_export_f32(self._h, capi.ptr(out, C.c_float)))
        return out

    def GenEmbeddingMap32(self):
        """word2vec.go:298-324: param rows narrowed to float32, keyed by word"""
        out = self.export_f32()
        return {w: out[i] for i, w in enumerate(self.dic.id2word)}

    def close(self):
        if self._h:
            capi.load().goctr_w2v_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def TrainEmbeddingIds(batches, capacity_words, window: int, dim: int, iter: int, seed=0, **kw) -> Word2Vec:
    """TrainEmbedding (wordemb.go:9-32) over batches of int64 item ids instead of a channel of their decimal strings;
    GenEmbeddingMap32's keys are then `mod.corpus.Dictionary()[0]`."""
    mod = Word2Vec(window=window, dim=dim, iter=iter, optimizer="hs", **kw)
    return mod.TrainIds(batches, capacity_words, seed)


def TrainEmbedding(inputCh, window: int, dim: int, iter: int, **kw) -> Word2Vec:
    """feature/embedding/wordemb.go:9-32: SkipGram + hierarchical softmax over an in-memory doc."""
    mod = Word2Vec(window=window, dim=dim, iter=iter, optimizer="hs", **kw)
    mod.Train(inputCh)
    return mod


def huffman_paths(counts, max_depth=100, want_ms=False):
    """dictionary/huffman.go:23-57 + node.go:39-42 on the host side of the library (goctr_huffman_build; needs no device):
    (path_off [V+1], inner-node ids, codes) of the root-to-leaf paths, the reference's tie-breaking."""
    L = capi.load()
    counts = np.ascontiguousarray(counts, np.int64)
    V = counts.size
    off = np.zeros(V + 1, np.int64)
    total, ms = C.c_int64(0), C.c_double(0)
    capi.check(L.goctr_huffman_build(capi.ptr(counts, C.c_int64), C.c_int64(V), C.c_int(max_depth), capi.ptr(off, C.c_int64), None, None,
                                     C.c_int64(0), C.byref(total), None))
    nodes = np.zeros(max(total.value, 1), np.int32)
    codes = np.zeros(max(total.value, 1), np.uint8)
    capi.check(L.goctr_huffman_build(capi.ptr(counts, C.c_int64), C.c_int64(V), C.c_int(max_depth), capi.ptr(off, C.c_int64),
                                     capi.ptr(nodes, C.c_int32), capi.ptr(codes, C.c_uint8), C.c_int64(total.value), C.byref(total), C.byref(ms)))
    res = (off, nodes[:total.value], codes[:total.value])
    return res + (ms.value,) if want_ms else res
