import numpy as np, os, sys
sys.path.insert(0, "/root/repo")
os.environ["GOCTR_NO_GRAPH"]="1"
from goctr_amd import capi, model as gm
import bench
emb, ub, it, uf, cf, y = bench.synth(1<<15, 42)
tab = gm.EmbeddingTable(emb); ds = gm.Dataset.ids(ub, it, uf, cf, y)
m = gm.DinNet(52,50,16,16,53); bench.init_weights(m,1)
cfg = capi.default_train_cfg(batch=8192, epochs=1)
gm.train_steps(m, ds, cfg, 20, emb=tab); capi.sync()
os.environ["GOCTR_DBG"]="chain,tn"
gm.train_steps(m, ds, cfg, 4, emb=tab); capi.sync()
