"""Chip / sliding-window inference (reference: ``instageo/model/infer_utils.py:57-136``).

The loop ``model(data) -> argmax(dim=1) -> int8`` (infer_utils.py:93-101) runs entirely on the device
(``ig_argmax_i8``); writing the per-chip rasters is outside the hot path (GeoTIFF needs rasterio), so
predictions are written as ``prediction_<name>.npy`` by the same 4-thread pool structure.
``sliding_window_inference`` is BASELINE.json configs[3]: a 10980^2 tile -> 49x49 windows of 224 (window rule of
``process_test``, dataloader.py:655-664), windows partitioned contiguously over ranks, final gather.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import distributed as D
from . import ops
from .dataloader import extract_windows, normalize_batch, window_origins


def save_prediction(prediction: np.ndarray, file_name: str, output_folder: str) -> str:
    """infer_utils.py:37-54 writes a GeoTIFF with the source profile; here: an .npy next to the same name."""
    base = os.path.splitext(os.path.basename(str(file_name)))[0]
    path = os.path.join(output_folder, f"prediction_{base}.npy")
    np.save(path, prediction)
    return path


def _engine_of(model):
    net = getattr(model, "net", model)
    return net, net.engine


@torch.no_grad()
def chip_inference(dataloader, output_folder: str, model, device: str = "gpu", num_workers: int = 4) -> Dict:
    """Run inference on chips and save one int8 class map per chip.  Returns {} (no carbon tracker here)."""
    os.makedirs(output_folder, exist_ok=True)
    net, eng = _engine_of(model)
    net.eval()
    with ThreadPoolExecutor(max_workers=num_workers) as executor:
        for (data, _), file_names in dataloader:
            data = data.to("cuda" if device == "gpu" else device)
            logits = eng.forward(data, training=False, save=False)
            if logits.shape[1] == 1:  # regression heads are out of scope; keep the reference's branch shape
                pred = logits.squeeze(1).cpu().numpy()
            else:
                pred = ops.argmax_i8(logits).cpu().numpy()
            futures = [executor.submit(save_prediction, p, f, output_folder) for p, f in zip(pred, file_names)]
            for fut in futures:
                fut.result()
    return {}


@torch.no_grad()
def sliding_window_inference(tile: torch.Tensor, model, mean: Sequence[float], std: Sequence[float], temporal_size: int = 1,
                             crop_size: int = 224, stride: int = 224, batch_size: int = 64,
                             constant_multiplier: Optional[float] = None, gather: bool = True
                             ) -> Tuple[Optional[torch.Tensor], List[Tuple[int, int]]]:
    """tile (T*C, S, S) int16|f32 on the device -> int8 class maps (n_windows, crop, crop) on rank 0.

    Every rank takes a contiguous block of the window list (no data-path collective), normalises its windows
    with ``ig_normalize_chips``, runs the forward pass and the fused argmax; ``gather`` collects the maps on
    rank 0 over RCCL.  Returns (maps or None on non-zero ranks, all window origins).
    """
    net, eng = _engine_of(model)
    net.eval()
    S = tile.shape[-1]
    origins = window_origins(S, crop_size, stride)
    world = D.world_size()
    rank = torch.distributed.get_rank() if world > 1 else 0
    lo, hi = D.shard_range(len(origins), rank, world)
    mine = origins[lo:hi]
    out = torch.empty((len(mine), crop_size, crop_size), dtype=torch.int8, device=tile.device)
    for i in range(0, len(mine), batch_size):
        chunk = mine[i : i + batch_size]
        x = normalize_batch(extract_windows(tile, chunk, crop_size), mean, std, temporal_size, constant_multiplier)
        logits = eng.forward(x, training=False, save=False)
        ops.argmax_i8(logits, out[i : i + len(chunk)])
    if gather and world > 1:
        counts = [D.shard_range(len(origins), r, world)[1] - D.shard_range(len(origins), r, world)[0] for r in range(world)]
        return D.gather_class_maps(out, counts, dst=0), origins
    return out, origins


def stitch_windows(maps: torch.Tensor, origins: Sequence[Tuple[int, int]], size: int, fill: int = -1) -> torch.Tensor:
    """Place non-overlapping window class maps back on a (size, size) int8 canvas (uncovered border = fill)."""
    crop = maps.shape[-1]
    canvas = torch.full((size, size), fill, dtype=torch.int8, device=maps.device)
    for m, (t, l) in zip(maps, origins):
        canvas[t : t + crop, l : l + crop] = m
    return canvas
