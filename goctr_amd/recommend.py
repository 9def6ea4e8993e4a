"""Host mirror of the parts of go-ctr's ``recommend`` package that sit either side of the hot path
(reference: recommend/rcmd.go).  Names and field meaning follow the Go types."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

# recommend/rcmd.go:19-28
SampleAssembler = 16
ItemEmbDim = 16
ItemEmbWindow = 5
UserBehaviorLen = 10


@dataclass
class SampleInfo:
    """recommend/rcmd.go:132-137"""
    UserProfileRange: tuple = (0, 0)
    UserBehaviorRange: tuple = (0, 0)
    ItemFeatureRange: tuple = (0, 0)
    CtxFeatureRange: tuple = (0, 0)

    def as_ranges(self) -> np.ndarray:
        return np.array([*self.UserProfileRange, *self.UserBehaviorRange, *self.ItemFeatureRange,
                         *self.CtxFeatureRange], np.int32)

    @staticmethod
    def from_dims(U: int, T: int, D: int, C: int) -> "SampleInfo":
        """the ranges GetSample records (rcmd.go:401-422)"""
        a, b, c = U, U + T * D, U + T * D + D
        return SampleInfo((0, a), (a, b), (b, c), (c, c + C))


@dataclass
class TrainSample:
    """recommend/rcmd.go:56-63: row-major X [Rows x XCols] float32, Y [Rows]"""
    X: np.ndarray
    Y: np.ndarray
    Rows: int
    XCols: int
    Info: SampleInfo = field(default_factory=SampleInfo)


class PredictAbstract:
    """recommend/rcmd.go:87-89"""

    def Predict(self, X: np.ndarray) -> np.ndarray:  # [n, XCols] -> [n, 1]
        raise NotImplementedError


class Fitter:
    """recommend/rcmd.go:95-97"""

    def Fit(self, sample: TrainSample) -> PredictAbstract:
        raise NotImplementedError
