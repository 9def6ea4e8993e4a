"""Host mirror of the reference's in-memory corpus for integer tokens, backed by the device.

Reference: feature/embedding/corpus/memory/memory.go (New :36, Load :76-102, IndexedDoc :53-62, Len :72),
corpus/dictionary/dictionary.go (Add :70-81, Len :39, IDFreq :64), corpus/cpsutil/cpsutil.go:58-78 (filters).
go-ctr's item2vec words are decimal item ids (example/movielens/feature.go:78), so the tokens are int64 here; they are
copied to HBM as they arrive and the dictionary / indexed doc are built there (csrc/corpus.hip)."""
import ctypes as C

import numpy as np

from . import capi


class Corpus:
    def __init__(self, capacity_words, min_count=5, max_count=-1):
        capi.init()
        self.min_count, self.max_count = min_count, max_count            # options.go:44-45 defaults
        self._h = C.c_void_p()
        capi.check(capi.load().goctr_corpus_create(C.c_int64(capacity_words), C.byref(self._h)))
        self._built = False

    # --- memory.go:76-102: the ItemSeqGenerator stream, batch by batch
    def append(self, keys):
        k = np.ascontiguousarray(keys, np.int64)
        capi.check(capi.load().goctr_corpus_append(self._h, capi.ptr(k, C.c_int64), C.c_int64(k.size)))
        self._built = False
        return self

    def Load(self, batches):
        for b in batches:
            self.append(b)
        return self.build()

    def build(self):
        capi.check(capi.load().goctr_corpus_build(self._h, C.c_int64(self.min_count), C.c_int64(self.max_count)))
        n, v, m = C.c_int64(), C.c_int64(), C.c_int64()
        capi.check(capi.load().goctr_corpus_info(self._h, C.byref(n), C.byref(v), C.byref(m)))
        self.n_words, self.V, self.n_indexed = n.value, v.value, m.value
        self._built = True
        return self

    def Len(self):                                                       # memory.go:72-74 (unfiltered)
        return self.n_words

    def Dictionary(self):
        """(id2key [V], cfs [V]): Dictionary.id2word / cfs"""
        id2key = np.empty(self.V, np.int64)
        cfs = np.empty(self.V, np.int64)
        capi.check(capi.load().goctr_corpus_get_dictionary(self._h, capi.ptr(id2key, C.c_int64), capi.ptr(cfs, C.c_int64)))
        return id2key, cfs

    def idoc(self):
        out = np.empty(self.n_words, np.int32)
        capi.check(capi.load().goctr_corpus_get_doc(self._h, capi.ptr(out, C.c_int32), None))
        return out

    def IndexedDoc(self):                                                # memory.go:53-62
        out = np.empty(self.n_indexed, np.int32)
        capi.check(capi.load().goctr_corpus_get_doc(self._h, None, capi.ptr(out, C.c_int32)))
        return out

    def close(self):
        if self._h:
            capi.load().goctr_corpus_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
