"""Chip / sliding-window inference (reference: ``instageo/model/infer_utils.py:37-136``).

The loop ``model(data) -> argmax(dim=1) -> int8`` (infer_utils.py:93-101) runs entirely on the device (``ig_argmax_i8``); the
per-chip rasters are written as ``prediction_*.tif`` with the source chip's georeferencing tags by the same 4-thread pool
structure (``save_prediction``, infer_utils.py:37-54, through :mod:`instageo_amd.tiff` instead of rasterio).

``sliding_window_inference`` is BASELINE.json configs[3]: a 10980^2 tile -> 49 x 49 windows of 224 (the window rule of
``process_test``, dataloader.py:655-664), gathered + normalised by ONE kernel launch per batch (``ig_normalize_windows``),
windows partitioned contiguously over ranks, final gather; ``stitch_windows`` puts the class maps back on the tile canvas
(overlapping windows: every pixel takes the window whose centre is nearest), ``tile_inference`` does file -> file.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import distributed as D
from . import ops, tiff
from .dataloader import gather_windows, origins_tensor, window_origins


def save_prediction(prediction: np.ndarray, file_name: str, output_folder: str, profile: Optional[Dict[str, Any]] = None) -> str:
    """Save one prediction as a TIFF next to the reference's naming (``chip`` -> ``prediction`` in the base name,
    infer_utils.py:51-54); ``profile`` = the source chip's profile (georeferencing tags are copied, count = 1)."""
    base = os.path.basename(str(file_name))
    out = base.replace("chip", "prediction") if "chip" in base else "prediction_" + base
    if not out.lower().endswith((".tif", ".tiff")):
        out = os.path.splitext(out)[0] + ".tif"
    path = os.path.join(output_folder, out)
    tiff.write(path, prediction, profile)
    return path


def _engine_of(model):
    net = getattr(model, "net", model)
    return net, net.engine


def _profile_of(file_name: str, dtype: np.dtype) -> Optional[Dict[str, Any]]:
    """infer_utils.py:103-113: the source profile with count=1 and the prediction dtype; None for in-memory chips."""
    if not (isinstance(file_name, str) and os.path.isfile(file_name) and file_name.lower().endswith((".tif", ".tiff"))):
        return None
    prof = dict(tiff.read_profile(file_name))
    prof.update(count=1, dtype=np.dtype(dtype).name)
    if np.dtype(dtype) == np.int8:
        prof["nodata"] = None  # the chip's NODATA value (-9999) does not exist in an int8 class map
        prof["tags"] = {k: v for k, v in prof["tags"].items() if k != 42113}
    return prof


@torch.no_grad()
def chip_inference(dataloader, output_folder: str, model, device: str = "gpu", num_workers: int = 4) -> Dict:
    """Run inference on chips and save one int8 class map (float32 for single-channel regression heads) per chip as
    ``prediction_*.tif``.  Returns {} (the reference returns CodeCarbon numbers; there is no tracker here)."""
    os.makedirs(output_folder, exist_ok=True)
    net, eng = _engine_of(model)
    net.eval()
    with ThreadPoolExecutor(max_workers=num_workers) as executor:
        for (data, _), file_names in dataloader:
            data = data.to("cuda" if device == "gpu" else device)
            logits = eng.forward(data, training=False, save=False)
            if logits.shape[1] == 1:  # regression (single output channel)
                pred = logits.squeeze(1).cpu().numpy()
            else:
                pred = ops.argmax_i8(logits).cpu().numpy()
            profiles = [_profile_of(f, pred.dtype) for f in file_names]
            futures = [executor.submit(save_prediction, p, f, output_folder, prof) for p, f, prof in zip(pred, file_names, profiles)]
            for fut in futures:
                fut.result()
    return {}


@torch.no_grad()
def sliding_window_inference(tile: torch.Tensor, model, mean: Sequence[float], std: Sequence[float], temporal_size: int = 1,
                             crop_size: int = 224, stride: int = 224, batch_size: int = 64,
                             constant_multiplier: Optional[float] = None, gather: bool = True
                             ) -> Tuple[Optional[torch.Tensor], List[Tuple[int, int]]]:
    """tile (T*C, S, S) int16|f32 on the device -> int8 class maps (n_windows, crop, crop) on rank 0.

    Every rank takes a contiguous block of the window list (no data-path collective); per batch ONE ``ig_normalize_windows``
    launch gathers and normalises its windows straight from the tile, then the forward pass and the fused argmax;
    ``gather`` collects the maps on rank 0 over RCCL.  Returns (maps or None on non-zero ranks, all window origins).
    """
    net, eng = _engine_of(model)
    net.eval()
    S = tile.shape[-1]
    origins = window_origins(S, crop_size, stride)
    world = D.world_size()
    rank = torch.distributed.get_rank() if world > 1 else 0
    lo, hi = D.shard_range(len(origins), rank, world)
    mine = origins_tensor(origins[lo:hi], tile.device)
    n = hi - lo
    out = torch.empty((n, crop_size, crop_size), dtype=torch.int8, device=tile.device)
    C = tile.shape[0] // temporal_size
    # balanced batches: ceil(n / batch_size) batches of nearly equal size (2401 windows at 108 per batch = 23 x 104-105, not 22 x 108 + 25:
    # a small ragged batch runs the persistent GEMM grids mostly empty and allocates a workspace of its own)
    nbatch = max(1, -(-n // batch_size))
    bs = -(-n // nbatch) if n else 1
    xbuf = torch.empty((max(bs, 1), C, temporal_size, crop_size, crop_size), dtype=torch.float32, device=tile.device)
    for i in range(0, n, bs):
        k = min(bs, n - i)
        x, _ = gather_windows(tile, mine[i : i + k], mean, std, temporal_size, crop_size, constant_multiplier, out=xbuf[:k])
        logits = eng.forward(x, training=False, save=False)
        ops.argmax_i8(logits, out[i : i + k])
    if gather and D.dp_active():
        counts = [D.shard_range(len(origins), r, world)[1] - D.shard_range(len(origins), r, world)[0] for r in range(world)]
        return D.gather_class_maps(out, counts, dst=0), origins
    return out, origins


def stitch_windows(maps: torch.Tensor, origins: Sequence[Tuple[int, int]], size, fill: int = -1) -> torch.Tensor:
    """Place window class maps back on a (H, W) int8 canvas; pixels no window covers (the remainder strip that the window rule
    drops, e.g. the last 4 px of a 10980 tile) = ``fill``.

    Overlap rule (stride < crop): a pixel takes the prediction of the window whose CENTRE is nearest in the Chebyshev metric
    (predictions are most reliable away from the window border); ties go to the earlier window in row-major order.  The result
    does not depend on the order in which windows are placed."""
    H, W = (size, size) if isinstance(size, int) else size
    crop = maps.shape[-1]
    canvas = torch.full((H, W), fill, dtype=torch.int8, device=maps.device)
    if len(origins) == 0:
        return canvas
    tops = sorted({t for t, _ in origins})
    lefts = sorted({l for _, l in origins})
    grid = len(tops) * len(lefts) == len(origins) and list(origins) == [(t, l) for t in tops for l in lefts]
    step_t = tops[1] - tops[0] if len(tops) > 1 else crop
    step_l = lefts[1] - lefts[0] if len(lefts) > 1 else crop
    regular = all(b - a == crop for a, b in zip(tops, tops[1:])) and all(b - a == crop for a, b in zip(lefts, lefts[1:]))
    if grid and step_t == crop and step_l == crop and regular:
        # non-overlapping regular grid: one strided copy
        ny, nx = len(tops), len(lefts)
        block = maps.view(ny, nx, crop, crop).permute(0, 2, 1, 3).reshape(ny * crop, nx * crop)
        canvas[tops[0] : tops[0] + ny * crop, lefts[0] : lefts[0] + nx * crop] = block
        return canvas
    ax = torch.arange(crop, device=maps.device, dtype=torch.float32) - (crop - 1) / 2.0
    dist = torch.maximum(ax.abs()[:, None], ax.abs()[None, :])  # Chebyshev distance to the window centre (half-integer grid)
    best = torch.full((H, W), float("inf"), dtype=torch.float32, device=maps.device)
    for m, (t, l) in zip(maps, origins):
        reg = best[t : t + crop, l : l + crop]
        take = dist < reg  # strict: ties keep the earlier window
        canvas[t : t + crop, l : l + crop][take] = m[take]
        reg[take] = dist[take]
    return canvas


@torch.no_grad()
def tile_inference(tile_path: str, output_folder: str, model, mean: Sequence[float], std: Sequence[float], temporal_size: int = 1,
                   crop_size: int = 224, stride: int = 224, batch_size: int = 64, constant_multiplier: Optional[float] = None,
                   no_data_value: Optional[float] = -9999, fill: int = -1, device: str = "cuda") -> Optional[str]:
    """GeoTIFF tile -> ``prediction_*.tif`` class map of the same georeferencing: read the (T*C, H, W)
    tile, sliding-window inference over all ranks, stitch, blank NODATA pixels (any band == ``no_data_value``) and uncovered
    border pixels with ``fill``, write on rank 0.  Returns the output path on rank 0, None elsewhere."""
    arr, profile = tiff.read(tile_path)
    if arr.shape[1] != arr.shape[2]:
        raise ValueError("tile_inference expects a square tile (the window rule of process_test uses one img_size)")
    t = torch.from_numpy(arr if arr.dtype in (np.int16, np.float32) else arr.astype(np.float32)).to(device)
    maps, origins = sliding_window_inference(t, model, mean, std, temporal_size, crop_size, stride, batch_size, constant_multiplier)
    if maps is None:
        return None
    canvas = stitch_windows(maps, origins, (arr.shape[1], arr.shape[2]), fill)
    if no_data_value is not None:
        canvas[(t == no_data_value).any(0)] = fill
    os.makedirs(output_folder, exist_ok=True)
    prof = dict(profile)
    prof.update(count=1, dtype="int8", nodata=fill)
    prof["tags"] = {k: v for k, v in profile["tags"].items() if k != 42113}
    return save_prediction(canvas.cpu().numpy(), tile_path, output_folder, prof)
