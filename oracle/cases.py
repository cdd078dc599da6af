"""Seeded parity cases shared by ``oracle/gen_golden.py`` (build container) and the tests (any box).

TEST INFRASTRUCTURE ONLY (see oracle/prithvi_oracle.py).  Inputs are regenerated from seeds, never stored.
"""
from __future__ import annotations

import numpy as np
import torch

from . import prithvi_oracle as O

CASES = {
    # name: (variant, T, ncls, B, depth)
    "tiny_t1_c2": ("prithvi_eo_tiny", 1, 2, 2, -1),
    "tiny_t3_c13": ("prithvi_eo_tiny", 3, 13, 2, -1),
    "v1_100_t1_c2": ("prithvi_eo_v1_100", 1, 2, 4, -1),  # BASELINE.json configs[0]
    "v1_100_t3_c13": ("prithvi_eo_v1_100", 3, 13, 1, -1),
    # the reference YAMLs' own per-GPU batches (train-mode BatchNorm couples the batch, so B = 4 / 1 do not cover them):
    "v1_100_t1_c2_b16": ("prithvi_eo_v1_100", 1, 2, 16, -1),  # configs/sen1floods11.yaml:13 (batch_size: 16)
    "v1_100_t3_c13_b8": ("prithvi_eo_v1_100", 3, 13, 8, -1),  # configs/multitemporal_crop_classification.yaml:14 (batch_size: 8)
    "v2_300_t1_c2": ("prithvi_eo_v2_300", 1, 2, 1, -1),  # BASELINE.json configs[4] architecture (D=1024, L=24, 16 heads)
    # the 600M shape family at depth 2 (model.py:154-177): D = 1280, 16 heads of 80, patch 14 (257 tokens), head kernels [5, 5, 5, 7]
    "v2_600_t1_c2": ("prithvi_eo_v2_600", 1, 2, 1, 2),
    # ... and at FULL depth (32 blocks, 630M parameters): eval logits only (an fp64 backward of it does not fit the build container)
    "v2_600_full_t1_c2": ("prithvi_eo_v2_600", 1, 2, 1, -1),
}
EVAL_ONLY = {"v2_600_full_t1_c2"}  # cases whose fixture holds the eval forward only

# instageo/model/configs/multitemporal_crop_classification.yaml:15-30
CROP_WEIGHTS = [0.386375, 0.661126, 0.548184, 0.640482, 0.876862, 0.925186, 3.249462,
                1.542289, 2.175141, 2.272419, 3.062762, 3.626097, 1.198702]  # fmt: skip

GRAD_KEYS = [
    "prithvi_encoder.cls_token",
    "prithvi_encoder.patch_embed.proj.weight",
    "prithvi_encoder.patch_embed.proj.bias",
    "prithvi_encoder.blocks.0.norm1.weight",
    "prithvi_encoder.blocks.0.attn.qkv.weight",
    "prithvi_encoder.blocks.0.attn.qkv.bias",
    "prithvi_encoder.blocks.0.attn.proj.weight",
    "prithvi_encoder.blocks.0.mlp.fc1.weight",
    "prithvi_encoder.blocks.0.mlp.fc2.bias",
    "prithvi_encoder.norm.bias",
    "segmentation_head.0.0.weight",
    "segmentation_head.0.2.weight",
    "segmentation_head.0.3.weight",
    "segmentation_head.3.0.bias",
    "segmentation_head.3.2.weight",
    "segmentation_head.5.weight",
    "segmentation_head.5.bias",
]


def case_config(name: str) -> O.OracleConfig:
    variant, T, ncls, _, depth = CASES[name]
    return O.make_config(variant, T, ncls, 224, depth)


def make_inputs(name: str, cfg: O.OracleConfig, B: int):
    """Seeded chips (already-normalised domain, as tests/model_tests/test_run.py:32-34) + labels with 5 % ignore."""
    seed = 1042 + sum(ord(c) for c in name)
    rng = np.random.default_rng(seed)
    img = rng.standard_normal((B, cfg.in_chans, cfg.num_frames, cfg.img_size, cfg.img_size)).astype(np.float32)
    lab = rng.integers(0, cfg.num_classes, size=(B, cfg.img_size, cfg.img_size)).astype(np.int64)
    lab[rng.random(lab.shape) < 0.05] = -1
    return torch.from_numpy(img), torch.from_numpy(lab)


def class_weights_for(ncls: int) -> torch.Tensor:
    return torch.tensor([1.0, 3.0]) if ncls == 2 else torch.tensor(CROP_WEIGHTS)


def sub(x: torch.Tensor, n: int = 4096) -> np.ndarray:
    """Deterministic strided subsample of a tensor (<= n values) -- the fixture sampling rule."""
    f = x.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].cpu().numpy().copy()


def make_stats_batches(case: str):
    """Seeded (data (B,C,T,H,W) f32 un-normalised reflectances, label (B,H,W) f32 with ignore value -1) batches for the
    mode=stats fixtures: ``t1`` = two batches of 3 single-date chips with 3 classes, ``t3`` = three batches of 2
    three-date chips with 5 classes.  Per-band offsets/scales differ so that a band mix-up shows."""
    T, nb, B, ncls = {"t1": (1, 2, 3, 3), "t3": (3, 3, 2, 5)}[case]
    g = torch.Generator().manual_seed(77 + T)
    out = []
    for _ in range(nb):
        scale = torch.linspace(0.02, 0.12, 6).view(1, 6, 1, 1, 1)
        shift = torch.linspace(0.05, 0.30, 6).view(1, 6, 1, 1, 1)
        data = torch.rand(B, 6, T, 32, 32, generator=g) * scale + shift
        label = torch.randint(-1, ncls, (B, 32, 32), generator=g).float()
        out.append((data, label))
    return out
