#!/usr/bin/env python3
"""Huffman build (goctr_huffman_build) timed at the vocabulary sizes of BASELINE configs[4]'s stress point and beyond, host-only
builder (GOCTR_HUFFMAN_DEVICE=0: round 3) against the build with the device (csrc/huffman.hip); writes
profiles/r04_huffman_timing.json.  build_ms = until the paths exist (host builder: in host vectors; device builder: resident in
HBM, where item2vec reads them), NOT the copy of the CSR out to the caller."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from goctr_amd import capi, embedding as ge  # noqa: E402

capi.init(0)
out = {"what": "goctr_huffman_build: Zipf(1.0) counts, max_depth 100, best of 3; build_ms = the tree + path build inside the library "
               "(device builder: paths resident in HBM), parts = GOCTR_HUFFMAN_PARTS of the last device run",
       "host_cores": os.cpu_count(), "rows": []}
for V in (10_681, 100_000, 1_000_000, 10_000_000):
    counts = np.maximum(1, (2e7 / np.arange(1, V + 1)).astype(np.int64))
    np.random.default_rng(1).shuffle(counts)
    row = {"V": V}
    for mode, tag in (("0", "host_only"), ("1", "with_device")):
        os.environ["GOCTR_HUFFMAN_DEVICE"] = mode
        best = None
        for _ in range(3):
            off, nodes, codes, ms = ge.huffman_paths(counts, want_ms=True)
            best = ms if best is None or ms < best else best
        row[tag + "_build_ms"] = round(best, 2)
        row["path_entries"] = int(nodes.size)
    os.environ.pop("GOCTR_HUFFMAN_DEVICE", None)
    out["rows"].append(row)
    print(row, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "huffman_timing.json"), "w"), indent=1)
