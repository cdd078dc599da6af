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

__all__ = ["RunningConfusionMatrix", "RunningAUC", "RunningRegressionMetrics", "metrics_from_matrix", "auc_from_histograms",
           "regression_metrics_from_sums"]


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


def auc_from_histograms(pos_hist: np.ndarray, neg_hist: np.ndarray) -> np.ndarray:
    """Per-class one-vs-rest ROC-AUC from score histograms, the arithmetic of metrics.py:238-248: positives outrank the
    negatives of lower bins, ties inside a bin count one half; NaN for a class without positives or without negatives."""
    pos = np.asarray(pos_hist, dtype=np.float64)
    neg = np.asarray(neg_hist, dtype=np.float64)
    cum_neg = np.cumsum(neg, axis=1) - neg  # negatives in strictly lower bins
    num = (pos * cum_neg + 0.5 * pos * neg).sum(axis=1)
    n_pos, n_neg = pos.sum(axis=1), neg.sum(axis=1)
    out = np.full(pos.shape[0], np.nan)
    ok = (n_pos > 0) & (n_neg > 0)
    out[ok] = num[ok] / (n_pos[ok] * n_neg[ok])
    return out


class RunningAUC:
    """Histogram-based streaming one-vs-rest ROC-AUC (reference: ``metrics.py:179-281``), device resident.

    ``update_from_logits`` bins softmax(logits) on the GPU (``ig_auc_update``; the reference walks the valid pixels in a
    Python loop, ``metrics.py:225-236``); ``update`` keeps the reference's (y_true, y_score) signature for host arrays.
    """

    def __init__(self, num_classes: int, n_bins: int = 1024, min_score: float = 0.0, max_score: float = 1.0,
                 ignore_index: Optional[int] = None, device: Optional[str] = None) -> None:
        self.num_classes, self.n_bins = num_classes, n_bins
        self.min_score, self.max_score = min_score, max_score
        self.ignore_index = ignore_index
        self._device = device
        self._hist: Optional[torch.Tensor] = None

    def device_hist(self, device=None) -> torch.Tensor:
        if self._hist is None:
            self._hist = torch.zeros(2, self.num_classes, self.n_bins, dtype=torch.int64, device=device or self._device or "cuda")
        return self._hist

    def update_from_logits(self, logits: torch.Tensor, labels: torch.Tensor) -> None:
        """logits (B, ncls, H, W) f32, labels (B, H, W) int64|int32|f32 on the device; ignored pixels are skipped."""
        if labels.dtype not in (torch.int64, torch.int32, torch.float32):
            labels = labels.long()
        ops.auc_update(logits.contiguous(), labels.contiguous(), self.ignore_index, self.device_hist(logits.device), self.n_bins,
                       self.min_score, self.max_score)

    def _bin(self, scores: np.ndarray) -> np.ndarray:
        s = np.minimum(self.max_score, np.maximum(self.min_score, scores))
        return ((s - self.min_score) / (self.max_score - self.min_score) * (self.n_bins - 1)).astype(np.int64)

    def update(self, y_true, y_score) -> None:
        """Reference signature: y_true (n,), y_score (n, C) probabilities (or (n,) positive-class scores when C == 2)."""
        y_true = np.asarray(y_true).ravel()
        y_score = np.asarray(y_score)
        if y_score.ndim == 1:
            if self.num_classes != 2:
                raise ValueError("For 1-D y_score, num_classes must be 2.")
            y_score = np.stack([1 - y_score, y_score], axis=1)
        if y_true.shape[0] != y_score.shape[0]:
            raise ValueError("y_true and y_score length mismatch.")
        if y_score.shape[1] != self.num_classes:
            raise ValueError("Second dim of y_score must equal num_classes.")
        add = np.zeros((2, self.num_classes, self.n_bins), dtype=np.int64)
        for c in range(self.num_classes):
            bins = self._bin(y_score[:, c])
            pos = y_true == c
            add[0, c] = np.bincount(bins[pos], minlength=self.n_bins)
            add[1, c] = np.bincount(bins[~pos], minlength=self.n_bins)
        h = self.device_hist()
        h += torch.from_numpy(add).to(h.device)

    @property
    def pos_hist(self) -> np.ndarray:
        return self.device_hist().cpu().numpy()[0]

    @property
    def neg_hist(self) -> np.ndarray:
        return self.device_hist().cpu().numpy()[1]

    def score(self, include_per_class: bool = True) -> dict:
        h = self.device_hist().cpu().numpy()
        per_class = auc_from_histograms(h[0], h[1])
        with np.errstate(all="ignore"):
            macro = np.nanmean(per_class) if np.isfinite(per_class).any() else float("nan")
        if include_per_class:
            return {"roc_auc_macro": macro, "roc_auc_per_class": per_class.tolist()}
        return {"roc_auc_macro": macro}

    def reset(self) -> None:
        if self._hist is not None:
            self._hist.zero_()


def regression_metrics_from_sums(sums, ee_bias: float = 0.05, ee_coef: float = 0.15, include_ee: bool = False) -> dict:
    """mae / rmse / r2 / pearson / ee_percentage from the streaming sums {n, Sx, Sy, Sxy, Sxx, Syy, S|e|, See, #EE}
    (x = truth, y = prediction), the formulas of metrics.py:354-420."""
    n, sx, sy, sxy, sxx, syy, sae, sse, nee = [float(v) for v in sums]
    nan = float("nan")
    mae = sae / n if n else nan
    rmse = float(np.sqrt(sse / n)) if n else nan
    r2 = pear = nan
    if n >= 2:
        xm, ym = sx / n, sy / n
        ss_tot = sxx - n * xm * xm
        r2 = nan if ss_tot == 0 else 1 - sse / ss_tot
        with np.errstate(invalid="ignore"):
            std_x, std_y = np.sqrt(sxx - n * xm * xm), np.sqrt(syy - n * ym * ym)
        pear = nan if std_x == 0 or std_y == 0 else float((sxy - n * xm * ym) / (std_x * std_y))
    return {"mae": mae, "rmse": rmse, "r2_score": r2, "pearson_corrcoef": pear,
            "ee_percentage": ((nee / n) * 100 if n else nan) if include_ee else None, "ee_bias": ee_bias, "ee_coef": ee_coef}  # fmt: skip


class RunningRegressionMetrics:
    """Streaming regression metrics (reference: ``metrics.py:288-420``); the nine running sums live on the device and are
    accumulated by ``ig_mse_loss`` (fused with the loss) or by :meth:`update` for host arrays."""

    def __init__(self, ee_bias: float = 0.05, ee_coef: float = 0.15, include_ee: bool = False, device: Optional[str] = None) -> None:
        self.ee_bias, self.ee_coef, self.include_ee = ee_bias, ee_coef, include_ee
        self._device = device
        self._sums: Optional[torch.Tensor] = None

    def device_sums(self, device=None) -> torch.Tensor:
        if self._sums is None:
            self._sums = torch.zeros(9, dtype=torch.float64, device=device or self._device or "cuda")
        return self._sums

    def update(self, y_true, y_pred) -> None:
        """Reference signature (host or device arrays of equal shape); no value is ignored here."""
        yt = torch.as_tensor(np.asarray(y_true) if not torch.is_tensor(y_true) else y_true).reshape(-1).float()
        yp = torch.as_tensor(np.asarray(y_pred) if not torch.is_tensor(y_pred) else y_pred).reshape(-1).float()
        if yt.shape != yp.shape:
            raise ValueError("y_true and y_pred shapes differ.")
        if yt.numel() == 0:
            return
        s = self.device_sums(yt.device if yt.is_cuda else None)
        scratch = torch.zeros(2, dtype=torch.float64, device=s.device)
        ops.mse_loss(yp.to(s.device).contiguous(), yt.to(s.device).contiguous(), float("nan"), False, scratch, None, s, self.ee_bias,
                     self.ee_coef, self.include_ee)

    @property
    def n(self) -> int:
        return int(self.device_sums()[0].item())

    def compute(self) -> dict:
        return regression_metrics_from_sums(self.device_sums().cpu().tolist(), self.ee_bias, self.ee_coef, self.include_ee)

    def reset(self) -> None:
        if self._sums is not None:
            self._sums.zero_()
