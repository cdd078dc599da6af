"""Streaming confusion-matrix metrics (reference: ``instageo/model/metrics.py:50-176``).

The k x k int64 matrix lives on the GPU and is updated by the HIP histogram kernels (either fused in
the loss kernel or through :func:`instageo_amd.ops.confusion_update`); only ``compute()`` copies the
k*k integers to the host, once per epoch, where the ratio arithmetic of the reference is reproduced.
This removes the reference's per-step ``.cpu().numpy()`` synchronisation (segmentation.py:143-151).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import ops

__all__ = ["RunningConfusionMatrix", "metrics_from_matrix"]


def _safe_div(num: np.ndarray, den: np.ndarray) -> np.ndarray:
    """Element-wise num / den, 0 where den == 0 (metrics.py:50-55)."""
    den = den.astype(float)
    out = np.zeros_like(den, dtype=float)
    np.divide(num, den, out=out, where=den != 0)
    return out


def metrics_from_matrix(matrix: np.ndarray, include_per_class: bool = True) -> dict:
    """accuracy + macro precision/recall/f1/jaccard (+ per-class lists) exactly as metrics.py:110-166."""
    matrix = np.asarray(matrix, dtype=np.int64)
    tp = np.diag(matrix)
    fp = matrix.sum(axis=0) - tp
    fn = matrix.sum(axis=1) - tp
    total = int(matrix.sum())
    prec = _safe_div(tp, tp + fp)
    rec = _safe_div(tp, tp + fn)
    f1 = _safe_div(2 * prec * rec, prec + rec)
    jac = _safe_div(tp, tp + fp + fn)
    out = {
        "accuracy": float("nan") if total == 0 else tp.sum() / total,
        "precision": prec.mean(),
        "recall": rec.mean(),
        "f1": f1.mean(),
        "jaccard": jac.mean(),
    }
    if include_per_class:
        out.update(
            precision_per_class=prec.tolist(), recall_per_class=rec.tolist(), f1_per_class=f1.tolist(),
            jaccard_per_class=jac.tolist(),
        )  # fmt: skip
    return out


class RunningConfusionMatrix:
    """Streaming confusion matrix for single-label classification, device resident."""

    def __init__(self, num_classes: int, ignore_index: Optional[int] = None, device: Optional[str] = None) -> None:
        self.num_classes = num_classes
        self.ignore_index = ignore_index
        self._device = device
        self._matrix: Optional[torch.Tensor] = None

    def device_matrix(self, device=None) -> torch.Tensor:
        """The int64 [k,k] device tensor that the HIP kernels accumulate into."""
        if self._matrix is None:
            dev = device or self._device or "cuda"
            self._matrix = torch.zeros(self.num_classes, self.num_classes, dtype=torch.int64, device=dev)
        return self._matrix

    def update(self, y_true, y_pred) -> None:
        """Add a (mini-)batch; accepts numpy arrays or tensors (moved to the GPU; no CPU fallback)."""
        yt = torch.as_tensor(np.asarray(y_true) if not torch.is_tensor(y_true) else y_true).reshape(-1).long()
        yp = torch.as_tensor(np.asarray(y_pred) if not torch.is_tensor(y_pred) else y_pred).reshape(-1).long()
        if yt.shape != yp.shape:
            raise ValueError("y_true and y_pred shapes differ.")
        if yt.numel() == 0:
            return
        m = self.device_matrix(yt.device if yt.is_cuda else None)
        ops.confusion_update(yt.to(m.device).contiguous(), yp.to(m.device).contiguous(), m, self.num_classes, self.ignore_index)

    @property
    def matrix(self) -> np.ndarray:
        if self._matrix is None:
            return np.zeros((self.num_classes, self.num_classes), dtype=np.int64)
        return self._matrix.cpu().numpy()

    @property
    def total(self) -> int:
        return int(self.matrix.sum())

    def compute(self, include_per_class: bool = True) -> dict:
        return metrics_from_matrix(self.matrix, include_per_class)

    def reset(self) -> None:
        if self._matrix is not None:
            self._matrix.zero_()
