import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from goctr_amd import capi, embedding as ge
capi.init(0)
V, dim, n = 10681, 16, 1_000_000
rng = np.random.default_rng(42)
p = 1.0 / np.arange(1, V + 1); p /= p.sum()
doc = rng.choice(V, size=n, p=p).astype(np.int32)
counts = np.bincount(doc, minlength=V) + 1
for streams in (2048, 8192, 16384, 32768, 65536, 131072):
    m = ge.Word2Vec(dim=dim, optimizer="hs", deterministic=False, streams=streams)
    m.create(counts); m.upload_doc(doc)
    m.train_resident(n * 6, lr=0.025); capi.sync()
    t0 = time.perf_counter()
    for _ in range(5): m.train_resident(n * 6, lr=0.025)
    capi.sync()
    dt = time.perf_counter() - t0
    print(streams, round(5 * n / dt / 1e6, 1), "M words/s")
