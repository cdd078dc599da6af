"""Dataset statistics of ``mode=stats`` on the device (reference: ``instageo/model/pipeline_utils.py:183-254``).

``compute_stats`` keeps the reference's definition -- per-band mean of the per-chip means and the square root of the mean of
the per-chip BIASED variances, class weights ``total / (n_present_classes * count)`` with the ignore value -1 dropped -- but
reduces every batch with two HIP kernels (``ig_chip_stats``: fp64 two-pass mean/variance per (chip, band);
``ig_label_hist``: label counts) instead of ``data.mean/var`` + ``np.unique`` on the host.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Iterable, List, Optional, Tuple

import torch

from . import ops
from .dataloader import eval_collate_fn, infer_collate_fn  # noqa: F401  (the reference keeps them here, pipeline_utils.py:77-104)

MAX_CLASSES = 255  # labels are counted in [-1, MAX_CLASSES)


def check_required_flags(required_flags: List[str], config: Any) -> None:
    """Raise when a required option still has its placeholder value "None" (pipeline_utils.py:43-54); ``config`` may be a dict
    (this package's run.py) or an attribute container (the reference's DictConfig)."""
    for flag_name in required_flags:
        value = config[flag_name] if isinstance(config, dict) else getattr(config, flag_name)
        if value == "None":
            raise RuntimeError(f"Flag --{flag_name} is required.")


def get_device() -> str:
    """"gpu" when a HIP device is visible, else "cpu" (pipeline_utils.py:57-74; there is no TPU / MPS path on this platform).
    The HIP kernels need "gpu": the product never falls back to a CPU implementation."""
    return "gpu" if torch.cuda.is_available() else "cpu"


def create_dataloader(dataset, batch_size: int, shuffle: bool = False, num_workers: Optional[int] = 1,
                      collate_fn: Optional[Callable] = None, pin_memory: bool = True) -> torch.utils.data.DataLoader:
    """``torch.utils.data.DataLoader`` with the reference's defaults (pipeline_utils.py:107-140).  Datasets of this package that
    hold their chips on the device (``ArrayChipDataset``, ``SyntheticChipDataset``) must stay in the main process
    (``num_workers=0``, no pinning): device tensors cannot cross a worker boundary."""
    num_workers = num_workers if num_workers is not None else 1
    on_device = bool(getattr(dataset, "device", None)) and str(getattr(dataset, "device")).startswith("cuda")
    if on_device:
        num_workers, pin_memory = 0, False
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, collate_fn=collate_fn,
                                       pin_memory=pin_memory)


def compute_class_weights(counts: Dict[int, int]) -> List[float]:
    """pipeline_utils.py:183-203."""
    total = sum(counts.values())
    ncls = len(counts)
    out = [0.0] * (int(max(counts.keys())) + 1)
    for cls, cnt in counts.items():
        out[int(cls)] = total / (ncls * cnt)
    return out


def compute_stats(data_loader: Iterable, is_reg_task: bool = False, device: str = "cuda") -> Tuple[List[float], List[float], Optional[List[float]]]:
    """``data_loader`` yields (data (B,C,T,H,W) or (B,C,H,W), label (B,H,W)); returns (mean, std, class_weights)."""
    sums = None
    hist = torch.zeros(MAX_CLASSES + 2, dtype=torch.int64, device=device)
    n = 0
    for data, label in data_loader:
        x = data.to(device=device, dtype=torch.float32).contiguous()
        if x.dim() == 4:
            x = x.unsqueeze(2)
        if sums is None:
            sums = torch.zeros(2 * x.shape[1], dtype=torch.float64, device=device)
        ops.chip_stats(x, sums)
        n += x.shape[0]
        if not is_reg_task:
            ops.label_hist(label.to(device=device, dtype=torch.float32).contiguous(), hist, lo=-1)
    if sums is None:
        raise ValueError("compute_stats: empty data loader")
    C = sums.numel() // 2
    mean = (sums[:C] / n).tolist()
    std = torch.sqrt(sums[C:] / n).tolist()
    weights = None
    if not is_reg_task:
        h = hist.cpu().tolist()
        if h[-1]:
            raise ValueError(f"compute_stats: {h[-1]} label pixels are not integers in [-1, {MAX_CLASSES})")
        counts = {v - 1: c for v, c in enumerate(h[:-1]) if c and v - 1 != -1}  # drop the ignore value -1
        weights = compute_class_weights(counts)
    return mean, std, weights
