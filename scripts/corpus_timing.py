"""Corpus load timing: 10^7 tokens, Zipf over 10^6 items; device build vs the C oracle on one core."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from goctr_amd import capi
from goctr_amd.corpus import Corpus
from oracle import pyoracle

n, vocab = 10_000_000, 1_000_000
rng = np.random.default_rng(0)
p = 1.0 / np.arange(1, vocab + 1) ** 1.05
keys = rng.choice(vocab, size=n, p=p / p.sum()).astype(np.int64)
capi.init()
for rep in range(3):
    c = Corpus(n)
    t0 = time.perf_counter(); c.append(keys); capi.sync(); t1 = time.perf_counter()
    c.build(); capi.sync(); t2 = time.perf_counter()
    print(f"device: upload {1e3*(t1-t0):.1f} ms, build {1e3*(t2-t1):.1f} ms, V={c.V}, indexed={c.n_indexed}")
    c.close()
t0 = time.perf_counter(); r = pyoracle.corpus_build(keys); t1 = time.perf_counter()
print(f"oracle (1 core, C): {1e3*(t1-t0):.1f} ms, V={r[1].size}")
