"""goctr_amd -- MI355X (gfx950) engine for go-ctr's CTR hot path.

Package layout (only what the path needs):
  csrc/            hand-written HIP kernels + the C-ABI (include/goctr.h) -> libgoctr_hip.so
  capi.py          ctypes binding of the C-ABI (the Python twin of the cgo stub in INTEGRATION.md)
  model.py         host mirror of go-ctr's model package: Train / Predict / InitForwardOnlyVm,
                   din.DinNet, youtube.YoutubeDnn incl. the Marshal JSON layout
  recommend.py     host mirror of recommend.SampleInfo / TrainSample / Fitter / PredictAbstract
  mlp.py           host mirror of nn.MLPClassifier behind model/mlp's Fit / Predict wrappers
  embedding.py     host mirror of feature/embedding.TrainEmbedding (item2vec)
"""
from . import capi  # noqa: F401
from .capi import GoctrError  # noqa: F401
