"""Chip dataset pieces of the hot path, device side (reference: ``instageo/model/dataloader.py``).

In scope: normalisation + tensor layout (``normalize_and_convert_to_tensor``
dataloader.py:495-524), random crop / flips hand-off (``process_and_augment`` :527-585), evaluation
window tiling (``process_test`` + ``crop_array`` :588-669) and the dataset item contract
(``InstaGeoDataset.__getitem__`` :875-902: ``(x:(C,T,H,W) f32, y:(H,W))``), the CSV-of-GeoTIFF dataset (host TIFF codec,
``tiff.py``) and the rotate / brightness / blur / noise augmentations (:144-386) as one ``ig_aug_*`` launch per batch.
The arithmetic (constant multiplier, mean/std normalisation) runs in ``ig_normalize_chips``; the training-time random
crop + flips + normalisation of a batch is ONE kernel (``ig_crop_flip_normalize``; the random draws stay on the host so
the reference's RNG stream can be replayed); window extraction is pure data movement on the device.
"""
from __future__ import annotations

import math
import random
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops


def _as_device_chip(x, device) -> torch.Tensor:
    t = torch.as_tensor(np.asarray(x)) if not torch.is_tensor(x) else x
    if t.dtype == torch.float64:
        t = t.float()
    if t.dtype not in (torch.int16, torch.float32):
        t = t.float()
    return t.to(device)


def normalize_and_convert_to_tensor(ims, label, mean: Sequence[float], std: Sequence[float], temporal_size: int = 1,
                                    constant_multiplier: Optional[float] = None, device: str = "cuda"):
    """(T*C,H,W) bands [band = t*C + c] -> (C,T,H,W) f32 ``(x - mean_c)/std_c`` ; label -> (H,W) tensor.

    ``ims`` may be an array/tensor (T*C,H,W) or a list of 2-D band arrays (the reference passes PIL images).
    """
    if isinstance(ims, (list, tuple)):
        ims = np.stack([np.asarray(i) for i in ims])
    x = _as_device_chip(ims, device)
    out = normalize_batch(x.unsqueeze(0), mean, std, temporal_size, constant_multiplier)[0]
    if label is not None:
        label = torch.as_tensor(np.asarray(label) if not torch.is_tensor(label) else label).squeeze().to(device)
    return out, label


def normalize_batch(x: torch.Tensor, mean: Sequence[float], std: Sequence[float], temporal_size: int = 1,
                    constant_multiplier: Optional[float] = None) -> torch.Tensor:
    """(B,T*C,H,W) int16|f32 device tensor -> (B,C,T,H,W) f32 normalised."""
    m = torch.as_tensor(mean, dtype=torch.float32, device=x.device)
    s = torch.as_tensor(std, dtype=torch.float32, device=x.device)
    return ops.normalize_chips(x.contiguous(), m, s, temporal_size, constant_multiplier)


def crop_array(arr, left: int, top: int, right: int, bottom: int):
    """dataloader.py:588-615 (2-D, 3-D or 4-D arrays/tensors)."""
    if arr.ndim == 2:
        return arr[top:bottom, left:right]
    if arr.ndim == 3:
        return arr[:, top:bottom, left:right]
    if arr.ndim == 4:
        return arr[:, :, top:bottom, left:right]
    raise ValueError("Input array must be a 2D, 3D or 4D array")


def window_origins(img_size: int, crop_size: int, stride: int) -> List[Tuple[int, int]]:
    """``for top in range(0, S-crop+1, stride): for left in ...`` (dataloader.py:655-664) -> [(top, left)]."""
    return [
        (top, left)
        for top in range(0, img_size - crop_size + 1, stride)
        for left in range(0, img_size - crop_size + 1, stride)
    ]


def extract_windows(tile: torch.Tensor, origins: Sequence[Tuple[int, int]], crop_size: int) -> torch.Tensor:
    """tile (T*C,S,S) [or label (S,S)] -> stacked windows (n, T*C, crop, crop) [(n, crop, crop)]: pure copies."""
    if tile.dim() == 2:
        return torch.stack([tile[t : t + crop_size, l : l + crop_size] for t, l in origins])
    return torch.stack([tile[:, t : t + crop_size, l : l + crop_size] for t, l in origins])


def origins_tensor(origins: Sequence[Tuple[int, int]], device) -> torch.Tensor:
    """(n, 2) int32 device tensor of (top, left) rows for ``ig_normalize_windows``."""
    return torch.tensor(list(origins), dtype=torch.int32).reshape(-1, 2).to(device)


def gather_windows(tile: torch.Tensor, origins, mean: Sequence[float], std: Sequence[float], temporal_size: int, crop_size: int,
                   constant_multiplier: Optional[float] = None, labels: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None):
    """Windows of ONE tile, gathered and normalised by a single kernel launch (replaces the per-window Python loop of
    ``crop_array`` + ``normalize_and_convert_to_tensor``): tile (T*C, S, S) int16|f32 on the device, ``origins`` a list of
    (top, left) or an (n, 2) int32 device tensor -> ((n, C, T, crop, crop) f32, labels (n, crop, crop) f32 or None)."""
    if not torch.is_tensor(origins):
        origins = origins_tensor(origins, tile.device)
    m = torch.as_tensor(mean, dtype=torch.float32, device=tile.device)
    s = torch.as_tensor(std, dtype=torch.float32, device=tile.device)
    lab = None if labels is None else labels.to(device=tile.device, dtype=torch.float32).contiguous()
    return ops.normalize_windows(tile.contiguous(), origins, m, s, temporal_size, crop_size, constant_multiplier, lab, out)


def process_test(x, y, mean: Sequence[float], std: Sequence[float], temporal_size: int = 1, img_size: int = 512,
                 crop_size: int = 224, stride: int = 224, constant_multiplier: Optional[float] = None, device: str = "cuda"):
    """Evaluation tiling (dataloader.py:618-669): -> (imgs (n,C,T,crop,crop) f32, labels (n,crop,crop))."""
    xt = _as_device_chip(x, device)
    yt = torch.as_tensor(np.asarray(y) if not torch.is_tensor(y) else y).to(device)
    origins = window_origins(img_size, crop_size, stride)
    if crop_size % 4 == 0 and yt.dim() == 2 and xt.shape[-2:] == yt.shape:
        imgs, labels = gather_windows(xt, origins, mean, std, temporal_size, crop_size, constant_multiplier, labels=yt)
        return imgs, (labels if yt.dtype == torch.float32 else labels.to(yt.dtype))
    imgs = normalize_batch(extract_windows(xt, origins, crop_size), mean, std, temporal_size, constant_multiplier)
    labels = extract_windows(yt, origins, crop_size)
    return imgs, labels


PHOTOMETRIC = ("rotate", "brightness", "blur", "noise")


def draw_augment_params(batch: int, src_hw: Tuple[int, int], im_size: int, crop: bool = True, augmentations: Optional[Dict] = None,
                        generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Host-side random draws of the data-movement part of ``process_and_augment`` for a batch: (B, 4) int32 rows
    (top, left, hflip, vflip).

    Crop origin as ``RandomCrop.get_params`` (uniform in [0, H - im], [0, W - im]; dataloader.py:73), each enabled flip
    with its probability ``p`` (dataloader.py:99, 131).  The other augmentations are drawn by :func:`draw_photometric_params`."""
    H, W = src_hw
    p = torch.zeros((batch, 4), dtype=torch.int32)
    if crop and (H > im_size or W > im_size):
        p[:, 0] = torch.randint(0, H - im_size + 1, (batch,), generator=generator, dtype=torch.int32)
        p[:, 1] = torch.randint(0, W - im_size + 1, (batch,), generator=generator, dtype=torch.int32)
    seen_other = False
    for name, cfg in (augmentations or {}).items():
        if not cfg.get("use", False):
            continue
        if name in PHOTOMETRIC:
            seen_other = True
            continue
        if name not in ("hflip", "vflip"):
            raise NotImplementedError(f"unknown augmentation {name!r} (the reference skips it with a warning, pipeline_utils.py:171-173)")
        if seen_other:  # the flips ride on the crop kernel, i.e. they run first; a rotation does not commute with them
            raise NotImplementedError("hflip / vflip must precede rotate / brightness / blur / noise in dataloader.augmentations")
        p[:, 2 if name == "hflip" else 3] = (torch.rand(batch, generator=generator) < cfg.get("p", 0.5)).to(torch.int32)
    return p


def rotate_coeffs(angle: float, size: int) -> Tuple[int, int, int, int, int, int]:
    """Pillow's ``Image.rotate`` (what ``transforms.functional.rotate`` calls for a PIL image, dataloader.py:183-186) as the six
    16.16 fixed-point coefficients of its nearest-neighbour affine walk: input x = (a2 + a1*y + a0*x) >> 16, input y =
    (a5 + a4*y + a3*x) >> 16 for output pixel (x, y) of a ``size`` x ``size`` image; ``ig_aug_rotate`` consumes them."""
    a = -math.radians(angle % 360.0)
    m0, m1, m3, m4 = round(math.cos(a), 15), round(math.sin(a), 15), round(-math.sin(a), 15), round(math.cos(a), 15)
    c = size / 2.0
    m2 = m0 * -c + m1 * -c + c
    m5 = m3 * -c + m4 * -c + c
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return fix(m0), fix(m1), fix(m2 + m0 * 0.5 + m1 * 0.5), fix(m3), fix(m4), fix(m5 + m3 * 0.5 + m4 * 0.5)


def gaussian_kernel2d(ksize: int, sigma: Sequence[float]) -> torch.Tensor:
    """torchvision's ``gaussian_blur`` kernel.  The reference hands its ``sigma_range`` pair to ``sigma=`` (dataloader.py:303-305),
    which torchvision reads as the fixed (sigma_x, sigma_y) -- there is no random draw; kept as is."""
    def k1(sig):
        half = (ksize - 1) * 0.5
        x = torch.linspace(-half, half, steps=ksize)
        pdf = torch.exp(-0.5 * (x / sig).pow(2))
        return pdf / pdf.sum()

    sx, sy = (float(sigma[0]), float(sigma[1])) if len(sigma) == 2 else (float(sigma[0]),) * 2
    return torch.mm(k1(sy)[:, None], k1(sx)[None, :])


def draw_photometric_params(batch: int, im_size: int, augmentations: Optional[Dict] = None, rng: Optional[random.Random] = None) -> List[Dict]:
    """Host-side draws of RandomRotation / RandomBrightnessContrast / RandomGaussianBlur / RandomGaussianNoise for a batch, in the
    order of the config (the reference composes them in that order, pipeline_utils.py:157-175): a list of
    ``{"name", per-chip parameter table, constants}``.  Like the reference the draws use Python's ``random`` (``random.random() < p``,
    then ``random.uniform`` for the parameters; dataloader.py:179-181, 228-229, 257, 330, 383) -- chip by chip, one augmentation
    after the other, so that a seeded ``random`` replays the reference's per-chip stream of these four."""
    rng = rng or random
    active = [(n, c) for n, c in (augmentations or {}).items() if c.get("use", False) and n in PHOTOMETRIC]
    if not active:
        return []
    tables: Dict[str, list] = {n: [] for n, _ in active}
    for _ in range(batch):
        for name, cfg in active:
            apply = rng.random() < cfg.get("p", 0.5)
            if name == "rotate":
                deg = float(cfg.get("degrees", 15))
                angle = rng.uniform(-deg, deg) if apply else 0.0
                tables[name].append([int(apply), *rotate_coeffs(angle, im_size), 0])
            elif name == "brightness":
                br, cr = cfg.get("brightness_range", (0.8, 1.2)), cfg.get("contrast_range", (0.8, 1.2))
                tables[name].append([float(apply), rng.uniform(*br), rng.uniform(*cr), 0.0] if apply else [0.0, 1.0, 1.0, 0.0])
            elif name == "blur":
                tables[name].append(int(apply))
            else:
                tables[name].append([int(apply), rng.getrandbits(31) if apply else 0])
    out = []
    for name, cfg in active:
        t = torch.tensor(tables[name], dtype=torch.float32 if name == "brightness" else torch.int32)
        out.append({"name": name, "table": t, "cfg": cfg})
    return out


def apply_photometric(buf: torch.Tensor, labels: Optional[torch.Tensor], plan: List[Dict], chip_no_data_value: float = 0.0,
                      label_no_data_value: float = -1.0, max_pixel_value: float = 10000.0, noise: Optional[torch.Tensor] = None):
    """Run a plan of :func:`draw_photometric_params` over a raw-domain batch (B, T*C, S, S) f32 [+ labels (B, S, S)]."""
    for step in plan:
        name, t, cfg = step["name"], step["table"].to(buf.device), step["cfg"]
        if name == "rotate":
            buf, labels = ops.aug_rotate(buf, t, chip_no_data_value, labels, label_no_data_value)
        elif name == "brightness":
            buf = ops.aug_brightness_contrast(buf, t, max_pixel_value)
        elif name == "blur":
            k2 = gaussian_kernel2d(int(cfg.get("kernel_size", 3)), cfg.get("sigma_range", (0.1, 2.0))).to(buf.device)
            buf = ops.aug_blur(buf, t, k2, max_pixel_value)
        else:
            buf = ops.aug_noise(buf, t, float(cfg.get("noise_std", 0.05)), max_pixel_value, noise)
    return buf, labels


def process_and_augment_batch(x: torch.Tensor, y: Optional[torch.Tensor], mean, std, temporal_size: int = 1, im_size: int = 224,
                              crop: bool = True, augmentations: Optional[Dict] = None, constant_multiplier: Optional[float] = None,
                              generator: Optional[torch.Generator] = None, params: Optional[torch.Tensor] = None,
                              label_no_data_value: float = -1.0, chip_no_data_value: float = 0.0, max_pixel_value: float = 10000.0,
                              rng: Optional[random.Random] = None, plan: Optional[List[Dict]] = None):
    """Batched ``process_and_augment`` on the device: x (B, T*C, Hs, Ws) int16|f32, y (B, Hs, Ws) or None ->
    ((B, C, T, im, im) f32 normalised, (B, im, im) f32).

    Crop + flips + normalise are ONE kernel (``ig_crop_flip_normalize``).  With rotate / brightness / blur / noise enabled the
    same kernel first writes the cropped, flipped chips in the raw domain (identity statistics), the ``ig_aug_*`` kernels run in
    config order, and ``ig_normalize_chips`` finishes (dataloader.py:570-585)."""
    B, TC, Hs, Ws = x.shape
    size = im_size if crop else Hs
    if not crop:
        assert Hs == Ws, "crop=False expects square chips"
    if params is None:
        params = draw_augment_params(B, (Hs, Ws), size, crop, augmentations, generator)
    if plan is None:
        plan = draw_photometric_params(B, size, augmentations, rng)
    m = torch.as_tensor(mean, dtype=torch.float32, device=x.device)
    s = torch.as_tensor(std, dtype=torch.float32, device=x.device)
    lab = None if y is None else y.to(device=x.device, dtype=torch.float32).contiguous()
    if not plan:
        return ops.crop_flip_normalize(x.contiguous(), params.to(x.device), m, s, temporal_size, size, constant_multiplier, lab)
    ident0, ident1 = torch.zeros(TC, device=x.device), torch.ones(TC, device=x.device)
    raw, lab = ops.crop_flip_normalize(x.contiguous(), params.to(x.device), ident0, ident1, 1, size, constant_multiplier, lab)
    raw, lab = apply_photometric(raw.view(B, TC, size, size), lab, plan, chip_no_data_value, label_no_data_value, max_pixel_value)
    return ops.normalize_chips(raw, m, s, temporal_size), lab


def process_and_augment(x, y, mean, std, temporal_size: int = 1, im_size: int = 224, crop: bool = True,
                        augmentations: Optional[Dict] = None, constant_multiplier: Optional[float] = None,
                        generator: Optional[torch.Generator] = None, device: str = "cuda"):
    """Random crop to ``im_size`` + optional h/v flips + normalise (dataloader.py:527-585), single chip."""
    xt = _as_device_chip(x, device).unsqueeze(0)
    yt = None if y is None else torch.as_tensor(np.asarray(y) if not torch.is_tensor(y) else y).to(device).reshape(1, *xt.shape[-2:])
    if not crop or (xt.shape[-2] <= im_size and xt.shape[-1] <= im_size):
        im_size, crop = xt.shape[-1], True  # nothing to crop: the kernel copies the whole chip (flips still apply)
    out, lab = process_and_augment_batch(xt, yt, mean, std, temporal_size, im_size, crop, augmentations, constant_multiplier, generator)
    return out[0], (None if lab is None else lab[0])


def process_label(label, replace_label: Optional[Sequence[float]] = None, reduce_to_zero: bool = False):
    """``np.where(y == replace_label[0], replace_label[1], y)`` then ``y -= 1`` (dataloader.py:742-746)."""
    y = label.clone() if torch.is_tensor(label) else np.array(label)
    if replace_label:
        if torch.is_tensor(y):
            y = torch.where(y == replace_label[0], torch.as_tensor(replace_label[1], dtype=y.dtype, device=y.device), y)
        else:
            y = np.where(y == replace_label[0], replace_label[1], y)
    if reduce_to_zero:
        y = y - 1
    return y


class SyntheticChipDataset(torch.utils.data.Dataset):
    """HLS-shaped synthetic chips generated on the device (there is no network for real data).

    Item contract of ``InstaGeoDataset`` (dataloader.py:875-902): ``(x:(C,T,H,W) f32 normalised, y:(H,W) f32)``.
    Raw domain: int16 uniform[0,10000), ``constant_multiplier`` 1e-4, 5 % of the label pixels = ignore_index.
    """

    def __init__(self, n: int, temporal: int, num_classes: int, mean, std, im_size: int = 224, ignore_index: int = -1,
                 constant_multiplier: Optional[float] = 1e-4, seed: int = 1042, device: str = "cuda", regression: bool = False):
        self.regression = regression  # labels = non-negative floats (e.g. aerosol optical depth) instead of class ids
        self.n, self.T, self.k = n, temporal, num_classes
        self.mean, self.std, self.S = list(mean), list(std), im_size
        self.ignore_index, self.mult, self.seed, self.device = ignore_index, constant_multiplier, seed, device

    def __len__(self) -> int:
        return self.n

    def raw(self, i: int) -> Tuple[torch.Tensor, torch.Tensor]:
        g = torch.Generator(device=self.device).manual_seed(self.seed + i)
        C = len(self.mean)
        x = torch.randint(0, 10000, (self.T * C, self.S, self.S), generator=g, device=self.device, dtype=torch.int16)
        if self.regression:
            y = torch.rand((self.S, self.S), generator=g, device=self.device) * 2.0
        else:
            y = torch.randint(0, self.k, (self.S, self.S), generator=g, device=self.device).float()
        y[torch.rand((self.S, self.S), generator=g, device=self.device) < 0.05] = float(self.ignore_index)
        return x, y

    def __getitem__(self, i: int):
        x, y = self.raw(i)
        return normalize_and_convert_to_tensor(x, y, self.mean, self.std, self.T, self.mult, self.device)

    def raw_batch(self, ids: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor]:
        """Un-normalised batch for the fused crop/flip/normalise kernel: (B, T*C, S, S) int16, (B, S, S) f32."""
        items = [self.raw(i) for i in ids]
        return torch.stack([a for a, _ in items]), torch.stack([b for _, b in items])


class ArrayChipDataset(torch.utils.data.Dataset):
    """Chips/labels held as arrays ``chips (N,T*C,H,W)``, ``labels (N,H,W)`` (stand-in for the GeoTIFF reader)."""

    def __init__(self, chips, labels, mean, std, temporal: int = 1, constant_multiplier: Optional[float] = None,
                 include_filenames: bool = False, names: Optional[List[str]] = None, device: str = "cuda",
                 replace_label: Optional[Sequence[float]] = None, reduce_to_zero: bool = False, no_data_value: Optional[float] = -9999):
        assert len(chips) == len(labels)
        self.no_data_value = no_data_value
        self.chips = chips
        # label clean-up of process_data (dataloader.py:742-746): value replacement, then shift to start from zero
        self.labels = [process_label(l, replace_label, reduce_to_zero) for l in labels] if (replace_label or reduce_to_zero) else labels
        self.mean, self.std, self.T, self.mult = list(mean), list(std), temporal, constant_multiplier
        self.include_filenames, self.names, self.device = include_filenames, names, device

    def __len__(self) -> int:
        return len(self.chips)

    def raw_batch(self, ids: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor]:
        """Un-normalised batch (B, T*C, H, W) int16|f32 and labels (B, H, W) f32 on the device."""
        x = torch.stack([_as_device_chip(self.chips[i], self.device) for i in ids])
        y = torch.stack([torch.as_tensor(np.asarray(self.labels[i])).float().squeeze() for i in ids]).to(self.device)
        return x, y

    def __getitem__(self, i: int):
        x, y = normalize_and_convert_to_tensor(self.chips[i], self.labels[i], self.mean, self.std, self.T, self.mult, self.device)
        if self.include_filenames:
            # third element: the NODATA mask of the chip, ``arr_x == no_data_value`` AFTER the constant multiplier exactly as the
            # reference computes it (dataloader.py:895-900; process_data has already scaled arr_x there)
            return (x, y), (self.names[i] if self.names else f"chip_{i:06d}"), nodata_mask(self.chips[i], self.no_data_value, self.mult)
        return x, y


def nodata_mask(chip, no_data_value: Optional[float], constant_multiplier: Optional[float] = None) -> np.ndarray:
    """``arr_x == no_data_value`` on the multiplier-scaled chip (T*C, H, W) -> bool array (dataloader.py:895-900)."""
    a = chip.cpu().numpy() if torch.is_tensor(chip) else np.asarray(chip)
    if no_data_value is None:
        return np.zeros(a.shape, dtype=bool)
    return (a * (1.0 if constant_multiplier is None else constant_multiplier)) == no_data_value


def get_raster_data(fname, is_label: bool = True, bands: Optional[List[int]] = None, no_data_value: Optional[int] = -9999,
                    mask_cloud: bool = True, water_mask: bool = False) -> np.ndarray:
    """All bands of a (Geo)TIFF as ``(count, H, W)`` (dataloader.py:672-704); ``bands`` selects image bands, label files are
    returned whole.  The reference's multi-file dict form (``open_mf_tiff_dataset``, data-creation side) is not supported."""
    from . import tiff

    if isinstance(fname, dict):
        raise NotImplementedError("multi-file tile dictionaries belong to the data-creation pipeline (SURVEY.md 2.1: out of scope)")
    data = tiff.read(fname)[0]
    if (not is_label) and bands:
        data = data[bands, ...]
    return data


def process_data(im_fname: str, mask_fname: Optional[str] = None, no_data_value: Optional[int] = -9999, reduce_to_zero: bool = False,
                 replace_label: Optional[Sequence[float]] = None, bands: Optional[List[int]] = None, constant_multiplier: float = 1.0,
                 mask_cloud: bool = False) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """Image (band selection, constant multiplier) and label (``replace_label``, ``reduce_to_zero``) arrays of one chip
    (dataloader.py:707-750)."""
    arr_x = get_raster_data(im_fname, is_label=False, bands=bands, no_data_value=no_data_value, mask_cloud=mask_cloud, water_mask=False)
    arr_x = arr_x * constant_multiplier
    arr_y = None
    if mask_fname:
        arr_y = process_label(get_raster_data(mask_fname), replace_label, reduce_to_zero)
    return arr_x, arr_y


def mask_label_with_chip(chips_path: str, labels_path: str, chip_no_data_value: int = 0, label_no_data_value: int = -1) -> bool:
    """True when the label has NO usable pixel: every pixel is NaN, the label NODATA value, or lies where the chip (first band
    of every time step) is NODATA (dataloader.py:753-783).  Only band 0 of every time step of the chip is decoded."""
    from . import tiff

    count = tiff.read_profile(chips_path)["count"]
    steps = max(1, count // 6)
    first = tiff.read(chips_path, bands=[6 * i for i in range(steps)])[0]
    has_data = np.where(first == chip_no_data_value, 0, 1).all(0)
    label = tiff.read(labels_path, bands=[0])[0][0]
    valid = (label != label_no_data_value) & (has_data == 1)
    if label.dtype.kind == "f":
        valid &= ~np.isnan(label)
    return not bool(np.any(valid))


def get_valid_filepaths(fname: str, input_root: str, no_data_value: int = -9999, ignore_index: int = -1) -> List[Tuple[str, Optional[str]]]:
    """(chip, label | None) path pairs of a CSV with an ``Input`` and optionally a ``Label`` column, paths relative to
    ``input_root``; rows whose chip is missing / unreadable, or whose label has no valid pixel over chip data, are dropped
    (dataloader.py:786-829).  Like the reference, a chip is only OPENED here (header parse) unless its label needs the data mask."""
    import os

    import pandas as pd

    from . import tiff

    data = pd.read_csv(fname)
    label_present = "Label" in data.columns
    out: List[Tuple[str, Optional[str]]] = []
    for _, row in data.iterrows():
        im = os.path.join(input_root, row["Input"])
        mk = os.path.join(input_root, row["Label"]) if label_present else None
        if not os.path.exists(im):
            continue
        try:
            if mk is None:
                tiff.read_profile(im)
                out.append((im, None))
            elif not mask_label_with_chip(im, mk, chip_no_data_value=no_data_value, label_no_data_value=ignore_index):
                out.append((im, mk))
        except Exception as exc:  # unreadable raster: the reference logs and skips
            print(f"[instageo_amd] skipping {im}: {exc}")
    print(f"Dropped a total of {len(data) - len(out)} rows")
    return out


class InstaGeoDataset(torch.utils.data.Dataset):
    """The reference's CSV-driven dataset (dataloader.py:832-906) on the TIFF codec of :mod:`instageo_amd.tiff`: ``filename`` is
    a CSV with an ``Input`` column (chip GeoTIFF, T*C bands) and optionally ``Label`` (segmentation map), paths relative to
    ``input_root``.  Items follow the reference contract: ``preprocess_func(arr_x, arr_y)``, and with ``include_filenames``
    the 3-tuple ``(preprocess_func(arr_x, arr_y), im_fname, arr_x == no_data_value)``.

    For the batched on-device loops of :mod:`instageo_amd.run` the dataset also offers the raw interface of
    ``ArrayChipDataset`` -- ``raw(i)`` / ``raw_batch(ids)`` (chips already multiplier-scaled by ``process_data``, hence
    ``mult = None``) and the ``mean`` / ``std`` / ``T`` / ``device`` attributes, taken from the keyword arguments or from the
    ``functools.partial`` that ``preprocess_func`` usually is."""

    def __init__(self, filename: str, input_root: str, preprocess_func, chip_no_data_value: Optional[float] = -9999,
                 label_no_data_value: Optional[float] = -1, replace_label=None, reduce_to_zero: bool = False,
                 constant_multiplier: float = 1.0, bands: Optional[List[int]] = None, include_filenames: bool = False,
                 mean: Optional[Sequence[float]] = None, std: Optional[Sequence[float]] = None, temporal_size: Optional[int] = None,
                 device: Optional[str] = None):
        self.input_root, self.preprocess_func, self.bands = input_root, preprocess_func, bands
        self.no_data_value, self.replace_label, self.reduce_to_zero = chip_no_data_value, replace_label, reduce_to_zero
        self.constant_multiplier, self.include_filenames = constant_multiplier, include_filenames
        self.file_paths = get_valid_filepaths(filename, input_root, chip_no_data_value, label_no_data_value)
        kw = getattr(preprocess_func, "keywords", None) or {}
        self.mean = list(mean if mean is not None else kw.get("mean", []))
        self.std = list(std if std is not None else kw.get("std", []))
        self.T = int(temporal_size if temporal_size is not None else kw.get("temporal_size", 1))
        self.mult = None  # process_data has applied the constant multiplier
        # items are produced by HIP kernels when the preprocessing runs on the device: such a dataset must stay in the main
        # process (pipeline_utils.create_dataloader forces num_workers=0 / no pinning from this attribute)
        self.device = str(device if device is not None else kw.get("device", "cpu"))

    def __len__(self) -> int:
        return len(self.file_paths)

    def _load(self, i: int) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        im_fname, mask_fname = self.file_paths[i]
        return process_data(im_fname, mask_fname, no_data_value=self.no_data_value, replace_label=self.replace_label,
                            reduce_to_zero=self.reduce_to_zero, bands=self.bands, constant_multiplier=self.constant_multiplier)

    def raw(self, i: int) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """Un-normalised chip (T*C, H, W), multiplier-scaled, and its label (H, W) | None (``process_data``)."""
        arr_x, arr_y = self._load(i)
        if arr_y is not None and arr_y.ndim == 3:
            arr_y = arr_y[0]
        return arr_x, arr_y

    def raw_batch(self, ids: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor]:
        """Un-normalised batch (B, T*C, H, W) f32 and labels (B, H, W) f32 on ``self.device`` (chips of one CSV share a size)."""
        xs, ys = zip(*[self.raw(i) for i in ids])
        if any(y is None for y in ys):
            raise RuntimeError("InstaGeoDataset.raw_batch needs a Label column (training / validation data)")
        x = torch.from_numpy(np.stack([np.asarray(a, dtype=np.float32) for a in xs])).to(self.device)
        y = torch.from_numpy(np.stack([np.asarray(a, dtype=np.float32) for a in ys])).to(self.device)
        return x, y

    def __getitem__(self, i: int):
        im_fname, _ = self.file_paths[i]
        arr_x, arr_y = self._load(i)
        if self.include_filenames:
            return self.preprocess_func(arr_x, arr_y), im_fname, arr_x == self.no_data_value
        return self.preprocess_func(arr_x, arr_y)


def eval_collate_fn(batch):
    """Concatenate windowed samples across the batch (pipeline_utils.py:78-89); items may carry the (filename, mask) tail."""
    items = [b[0] if isinstance(b[0], tuple) else b for b in batch]
    return torch.cat([it[0] for it in items], 0), torch.cat([it[1] for it in items], 0)


def infer_collate_fn(batch):
    """((data, labels), filenames) batches (pipeline_utils.py:92-104); the NODATA mask (third element) is not collated."""
    data = torch.stack([b[0][0] for b in batch])
    labels = [b[0][1] for b in batch]
    return (data, labels), [b[1] for b in batch]
