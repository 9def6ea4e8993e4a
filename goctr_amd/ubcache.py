"""ubcache -- host mirror of go-ctr's user-behaviour cache (feature/ubcache/cache.go) with the lookups on the device.

    TimeSeq                     cache.go:8-12      (sequence in timestamp-descending order)
    UserBehaviorCache           cache.go:16-68     Set / BatchSet / Delete / Clear / Get
    TimeSeq.Filter              cache.go:71-94

Set / BatchSet / Delete / Clear edit a host dictionary; the CSR image in HBM is rebuilt lazily on the next lookup.
`Get` answers one key like the reference; `get_batch` answers many keys per call (goctr_ubcache_get) and
`model.Dataset.keys` assembles a whole training set on the device (goctr_dataset_create_keys) -- the replacement for
the per-sample gather of recommend.GetSample / GetSampleVector (rcmd.go:339-536).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import capi


@dataclass
class TimeSeq:
    """cache.go:8-12"""
    Ts: list = field(default_factory=list)
    Items: list = field(default_factory=list)


class UserBehaviorCache:
    def __init__(self):
        self.ub = {}
        self._h = None
        self._users = None      # user id -> dense row of the CSR

    # ---- cache.go:22-56
    def Set(self, userId, seq: TimeSeq):
        self.ub[int(userId)] = seq
        self._drop()

    def BatchSet(self, ub: dict):
        for k, v in ub.items():
            self.ub[int(k)] = v
        self._drop()

    def Delete(self, userId):
        self.ub.pop(int(userId), None)
        self._drop()

    def Clear(self):
        self.ub = {}
        self._drop()

    # ---- device image
    def _drop(self):
        if self._h:
            capi.load().goctr_ubcache_destroy(self._h)
        self._h = None

    def user_index(self):
        """dense index of every cached user id (order of the CSR rows)"""
        if self._users is None or self._h is None:
            self.device()
        return self._users

    def device(self):
        if self._h is None:
            capi.init()
            ids = sorted(self.ub)
            if not ids:
                raise KeyError("the behaviour cache is empty")
            self._users = {u: i for i, u in enumerate(ids)}
            off = np.zeros(len(ids) + 1, np.int64)
            for i, u in enumerate(ids):
                off[i + 1] = off[i] + len(self.ub[u].Ts)
            items = np.concatenate([np.asarray(self.ub[u].Items, np.int32) for u in ids]) if off[-1] else np.zeros(0, np.int32)
            ts = np.concatenate([np.asarray(self.ub[u].Ts, np.int64) for u in ids]) if off[-1] else np.zeros(0, np.int64)
            self._h = C.c_void_p()
            capi.check(capi.load().goctr_ubcache_create(C.c_int64(len(ids)), capi.ptr(off, C.c_int64),
                                                        capi.ptr(np.ascontiguousarray(items), C.c_int32),
                                                        capi.ptr(np.ascontiguousarray(ts), C.c_int64), C.byref(self._h)))
        return self._h

    # ---- lookups
    def get_batch(self, userIds, maxTs, count):
        """ids [n, count] int32, -1 = empty slot; users missing from the cache raise KeyError like Get's error"""
        h = self.device()
        idx = np.array([self._users[int(u)] for u in userIds], np.int32)      # KeyError = "user %d not found"
        ts = np.ascontiguousarray(maxTs, np.int64)
        out = np.empty((idx.size, count), np.int32)
        capi.check(capi.load().goctr_ubcache_get(h, capi.ptr(idx, C.c_int32), capi.ptr(ts, C.c_int64), C.c_int64(idx.size),
                                                 C.c_int(count), capi.ptr(out, C.c_int32)))
        return out

    def Get(self, userId, maxTs, count) -> TimeSeq:
        """cache.go:58-68.  count == 0 means "all" (cache.go:76-78)."""
        if int(userId) not in self.ub:
            raise KeyError(f"user {userId} not found")
        seq = self.ub[int(userId)]
        n = count if count else len(seq.Ts)
        if n == 0:
            return TimeSeq([], [])
        ids = self.get_batch([userId], [maxTs], n)[0]
        k = int((ids >= 0).sum()) if (ids < 0).any() else n
        # the timestamps ride along on the host (the device returns the item ids, which is what the model consumes)
        ts = np.asarray(seq.Ts, np.int64)
        mts = ts[0] if maxTs == 0 else maxTs
        first = int(np.argmax(ts <= mts)) if (ts <= mts).any() else len(ts)
        return TimeSeq(ts[first:first + k].tolist(), [int(x) for x in ids[:k]])

    def __del__(self):
        try:
            self._drop()
        except Exception:
            pass


def NewUserBehaviorCache():
    return UserBehaviorCache()
