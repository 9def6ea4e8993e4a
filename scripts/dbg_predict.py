"""s_memtime phase stamps of the forward-only chain kernel (workgroup 0) for a 16384-row predict launch"""
import numpy as np, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GOCTR_NO_GRAPH"] = "1"
from goctr_amd import capi, model as gm
import bench
emb, ub, it, uf, cf, y = bench.synth(1 << 16, 42)
tab = gm.EmbeddingTable(emb); ds = gm.Dataset.ids(ub, it, uf, cf, y)
m = gm.DinNet(52, 50, 16, 16, 53); bench.init_weights(m, 1)
for _ in range(3):
    gm.predict_dataset(m, ds, 4096, emb=tab)
capi.sync()
os.environ["GOCTR_DBG"] = "chain"
gm.predict_dataset(m, ds, 4096, emb=tab); capi.sync()
