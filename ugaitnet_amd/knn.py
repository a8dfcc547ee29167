"""GPU k-nearest-neighbour classifier over gait signatures: the subset of sklearn's KNeighborsClassifier interface the
reference's evaluation main uses (mains/mj_testUWYHGaitNet_open_tum.py:328-341: `KNeighborsClassifier(n_neighbors=knn)`,
`.fit(codes, labels)`, `.predict(codes)`), executed by libugaitnet_hip.so (ugn_knn_predict).  No CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr

__all__ = ["KNeighborsClassifier"]


def _device():
    if not torch.cuda.is_available():
        raise _lib.UgnError("ugaitnet_amd.knn needs a GPU (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _codes(x, dev):
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))
    return t.to(device=dev, dtype=torch.float32).contiguous()


class KNeighborsClassifier:
    """Euclidean metric, uniform weights, majority vote; a tied vote goes to the smallest label (as sklearn's)."""

    def __init__(self, n_neighbors=5):
        if not 1 <= int(n_neighbors) <= 16:
            raise ValueError("n_neighbors must be in 1..16")
        self.n_neighbors = int(n_neighbors)
        self._x = None

    def fit(self, X, y):
        dev = _device()
        self._x = _codes(X, dev)
        if self._x.dim() != 2:
            raise ValueError("X must be [n_samples, n_features]")
        yv = np.asarray(y.cpu() if isinstance(y, torch.Tensor) else y).reshape(-1)
        if yv.shape[0] != self._x.shape[0]:
            raise ValueError("X and y disagree on the number of samples")
        # labels may be any sortable values (the reference passes float arrays): classify over their sorted unique set
        self.classes_, inv = np.unique(yv, return_inverse=True)
        self._y = torch.from_numpy(inv.astype(np.int32)).to(dev)
        if self.n_neighbors > self._x.shape[0]:
            raise ValueError("n_neighbors > n_samples")
        return self

    def kneighbors_and_predict(self, X):
        if self._x is None:
            raise RuntimeError("fit() first")
        q = _codes(X, self._x.device)
        if q.dim() != 2 or q.shape[1] != self._x.shape[1]:
            raise ValueError("X must be [n_queries, %d]" % self._x.shape[1])
        nq, ng, d, k = q.shape[0], self._x.shape[0], q.shape[1], self.n_neighbors
        pred = torch.empty((nq,), dtype=torch.int32, device=q.device)
        nbr = torch.empty((nq, k), dtype=torch.int32, device=q.device)
        nbytes = _lib.load().ugn_knn_ws(ng, nq)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=q.device)
        call("ugn_knn_predict", ptr(self._x), ptr(self._y), ptr(q), ng, nq, d, k, ptr(pred), ptr(nbr), ptr(ws), nbytes,
             C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return nbr.cpu().numpy(), self.classes_[pred.cpu().numpy()]

    def predict(self, X):
        return self.kneighbors_and_predict(X)[1]

    def score(self, X, y):
        return float(np.mean(self.predict(X) == np.asarray(y).reshape(-1)))
