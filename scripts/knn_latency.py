#!/usr/bin/env python3
"""Searcher.Search as the reference calls it -- ONE query per call (search.go:92-134) -- and in blocks: latency per call of
goctr_searcher_search over V = 10^6 x 16 float64 items, scan path (default) vs tile path (GOCTR_KNN_SCAN=0).
usage: [KNN_LATENCY_Q=1,8,64,256] [KNN_LATENCY_SCAN=1,0] python scripts/knn_latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goctr_amd import capi, search as gs

capi.init(0)
V, D, k = 1_000_000, 16, 10
rng = np.random.default_rng(42)
items = rng.standard_normal((V, D))
s = gs.Searcher([""] * V, items)
QS = tuple(int(x) for x in os.environ.get("KNN_LATENCY_Q", "1,8,64,256").split(","))
for scan in os.environ.get("KNN_LATENCY_SCAN", "1,0").split(","):
    os.environ["GOCTR_KNN_SCAN"] = scan
    for Q in QS:
        q = rng.standard_normal((Q, D))
        for _ in range(5):
            s.search_vectors(q, k)
        n = 200 if Q <= 64 else 50
        t = []
        for _ in range(n):
            t0 = time.perf_counter(); s.search_vectors(q, k); t.append(time.perf_counter() - t0)
        t = np.sort(np.array(t)) * 1e6
        print(f"GOCTR_KNN_SCAN={scan} Q {Q:4d}: p50 {t[len(t)//2]:8.1f} us  p99 {t[int(len(t)*0.99)-1]:8.1f} us  per query {t[len(t)//2]/Q:7.2f} us  ({Q/(t[len(t)//2]*1e-6):.0f} queries/s)")
