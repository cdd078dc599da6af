"""CPU restatement of the reference's photometric / resampling augmentations (TEST INFRASTRUCTURE ONLY -- imported by
``tests/`` as the checker of the ``ig_aug_*`` kernels; the product never imports it).

Reference: ``instageo/model/dataloader.py:144-386`` (RandomRotation, RandomBrightnessContrast, RandomGaussianBlur,
RandomGaussianNoise).  The resampling / filtering itself lives in third-party code the reference calls:

* ``torchvision.transforms.functional.rotate`` on a PIL image (torchvision 0.23.0, ``uv.lock:1922``; absent from this image) is
  ``PIL.Image.rotate(angle, resample=NEAREST, expand=False, fillcolor=fill)``.  Pillow IS in this image (12.2.0):
  :func:`rotate_nearest` restates Pillow's matrix construction (``Image.rotate``) and its 16.16 fixed-point nearest-neighbour
  affine walk (``Geometry.c: affine_fixed``) and ``tests/test_cpu_augment.py`` pins it bit for bit against ``PIL.Image.rotate``.
* ``torchvision.transforms.functional.gaussian_blur`` on a tensor: separable kernel ``exp(-0.5 (x / sigma)^2)`` normalised to sum 1
  on ``linspace(-(k-1)/2, (k-1)/2, k)``, reflect padding, depth-wise ``conv2d``.  The reference passes its ``sigma_range`` tuple as
  ``sigma`` (dataloader.py:303-305), which torchvision reads as the FIXED pair (sigma_x, sigma_y) = (0.1, 2.0) -- no random draw;
  restated as such.  Parity unpinned by any reference test (``test_dataloader.py`` checks shapes and value ranges only).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def rotate_matrix(angle: float, w: int, h: int) -> Tuple[float, float, float, float, float, float]:
    """Pillow's ``Image.rotate`` inverse affine matrix (output pixel -> input position) about the image centre."""
    angle = angle % 360.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    cx, cy = w / 2.0, h / 2.0
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return tuple(m)


def rotate_fixed_coeffs(angle: float, w: int, h: int) -> Tuple[int, int, int, int, int, int]:
    """The six 16.16 fixed-point coefficients of Pillow's ``affine_fixed`` walk (pixel centres folded into a2 / a5)."""
    m = rotate_matrix(angle, w, h)
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


def rotate_nearest(arr: np.ndarray, angle: float, fill: float) -> np.ndarray:
    """``transforms.functional.rotate(Image.fromarray(arr), angle, fill=fill)`` for one 2-D band (dataloader.py:183-186)."""
    h, w = arr.shape
    ang = angle % 360.0
    if ang == 0.0:
        return arr.copy()
    if ang == 180.0:
        return arr[::-1, ::-1].copy()
    if ang in (90.0, 270.0) and w == h:
        return np.rot90(arr, 1 if ang == 90.0 else 3).copy()
    a0, a1, a2, a3, a4, a5 = rotate_fixed_coeffs(angle, w, h)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.int64), np.arange(w, dtype=np.int64), indexing="ij")
    xin = (a2 + a1 * ys + a0 * xs) >> 16
    yin = (a5 + a4 * ys + a3 * xs) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.full_like(arr, fill)
    out[ok] = arr[yin[ok], xin[ok]]
    return out


def brightness_contrast(arr: np.ndarray, bright: float, contrast: float, max_pixel: float) -> np.ndarray:
    """One band of ``random_brightness_contrast`` (dataloader.py:230-238), float32 like the reference."""
    t = torch.from_numpy(np.asarray(arr, dtype=np.float32)) * bright
    mean = t.mean()
    t = (t - mean) * contrast + mean
    return t.clamp_(0, max_pixel).numpy()


def gaussian_kernel1d(ksize: int, sigma: float) -> torch.Tensor:
    half = (ksize - 1) * 0.5
    x = torch.linspace(-half, half, steps=ksize)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


def gaussian_blur(arr: np.ndarray, ksize: int, sigma: Sequence[float], max_pixel: float) -> np.ndarray:
    """One band of ``add_gaussian_blur`` (dataloader.py:296-314): clip, scale to [0, 1], blur, clamp, rescale, uint16 cast."""
    a = np.clip(np.asarray(arr, dtype=np.float32), 0, max_pixel) / np.float32(max_pixel)
    t = torch.from_numpy(a)[None, None]
    kx, ky = gaussian_kernel1d(ksize, float(sigma[0])), gaussian_kernel1d(ksize, float(sigma[1]))
    k2 = torch.mm(ky[:, None], kx[None, :])
    p = ksize // 2
    t = F.conv2d(F.pad(t, [p, p, p, p], mode="reflect"), k2[None, None])
    t = torch.clamp(t, 0.0, 1.0)[0, 0] * max_pixel
    return t.numpy().astype(np.uint16)


def gaussian_noise(arr: np.ndarray, noise: np.ndarray, noise_std: float, max_pixel: float) -> np.ndarray:
    """One band of ``add_gaussian_noise`` (dataloader.py:356-368) with the standard-normal field given."""
    a = np.clip(np.asarray(arr, dtype=np.float32), 0, max_pixel) / np.float32(max_pixel)
    t = torch.from_numpy(a) + torch.from_numpy(np.asarray(noise, dtype=np.float32)) * noise_std
    t = torch.clamp(t, 0.0, 1.0) * max_pixel
    return t.numpy().astype(np.uint16)
