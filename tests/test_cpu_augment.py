"""Oracle of the rotate / brightness / blur / noise augmentations (oracle/augment_oracle.py) and the host side of their device
path: pinned against Pillow (the library the reference's RandomRotation calls through torchvision) -- no GPU needed."""
import math
import random

import numpy as np
import pytest
import torch

from oracle import augment_oracle as AO
from instageo_amd import dataloader as DL

PIL = pytest.importorskip("PIL.Image")


@pytest.mark.parametrize("shape", [(224, 224), (32, 32), (17, 23)])
def test_rotation_oracle_is_pillow_bit_for_bit(shape):
    rng = np.random.default_rng(shape[0])
    arr = rng.uniform(0, 10000, shape).astype(np.float32)
    angles = [0, 90, 180, 270, -90, 5.3, -10, 10, 45, -45, 123.456, 359.9, 1e-3, -1e-3] + list(rng.uniform(-15, 15, 12))
    for ang in angles:
        ref = np.array(PIL.fromarray(arr).rotate(ang, resample=PIL.NEAREST, expand=False, fillcolor=-1))
        assert np.array_equal(AO.rotate_nearest(arr, ang, -1), ref), f"angle {ang}"


def test_host_rotate_coeffs_equal_the_oracle():
    for ang in (0.0, 3.7, -14.2, 90.0, 180.0, 271.3):
        assert DL.rotate_coeffs(ang, 224) == AO.rotate_fixed_coeffs(ang, 224, 224)


def test_gaussian_kernel_matches_the_oracle_and_sums_to_one():
    k2 = DL.gaussian_kernel2d(3, (0.1, 2.0))
    kx, ky = AO.gaussian_kernel1d(3, 0.1), AO.gaussian_kernel1d(3, 2.0)
    assert torch.equal(k2, torch.mm(ky[:, None], kx[None, :]))
    assert abs(float(k2.sum()) - 1.0) < 1e-6
    # sigma_x = 0.1: practically no blur along x, sigma_y = 2.0: ~box along y (the reference's fixed pair, dataloader.py:303-305)
    assert k2[1, 1] > 0.3 and k2[1, 0] < 1e-6 and k2[0, 1] > 0.3


def test_blur_and_noise_oracle_known_answers():
    flat = np.full((8, 8), 5000.0, np.float32)
    assert np.all(AO.gaussian_blur(flat, 3, (0.1, 2.0), 10000.0) == 5000)  # a constant image stays constant (reflect padding)
    z = np.zeros((8, 8), np.float32)
    assert np.all(AO.gaussian_noise(flat, z, 0.05, 10000.0) == 5000)
    assert np.all(AO.gaussian_noise(flat, z + 1.0, 0.05, 10000.0) == 5500)
    assert np.all(AO.gaussian_noise(flat, z - 100.0, 0.05, 10000.0) == 0) and np.all(AO.gaussian_noise(flat, z + 100.0, 0.05, 10000.0) == 10000)
    bc = AO.brightness_contrast(np.array([[1000.0, 3000.0]], np.float32), 1.0, 2.0, 10000.0)
    assert np.allclose(bc, [[0.0, 4000.0]])  # contrast doubles the distance to the mean (2000), clamp at 0


def test_photometric_draws_follow_the_config_order_and_python_random():
    augs = {"hflip": {"use": True, "p": 0.5}, "rotate": {"use": True, "p": 0.5, "degrees": 10},
            "brightness": {"use": True, "p": 0.5, "brightness_range": [0.8, 1.2], "contrast_range": [0.9, 1.1]},
            "blur": {"use": False, "p": 0.5}, "noise": {"use": True, "p": 1.0, "noise_std": 0.05}}
    plan = DL.draw_photometric_params(5, 224, augs, random.Random(3))
    assert [s["name"] for s in plan] == ["rotate", "brightness", "noise"]
    # replay the reference's per-chip stream: random() < p, then the uniform draws, augmentation after augmentation
    r = random.Random(3)
    for b in range(5):
        rot = r.random() < 0.5
        ang = r.uniform(-10, 10) if rot else 0.0
        row = plan[0]["table"][b].tolist()
        assert row[0] == int(rot) and tuple(row[1:7]) == DL.rotate_coeffs(ang, 224)
        br = r.random() < 0.5
        exp = [1.0, r.uniform(0.8, 1.2), r.uniform(0.9, 1.1)] if br else [0.0, 1.0, 1.0]
        assert np.allclose(plan[1]["table"][b, :3].numpy(), exp, rtol=1e-6)
        assert r.random() < 1.0
        assert plan[2]["table"][b, 0] == 1 and plan[2]["table"][b, 1] == r.getrandbits(31)
    assert DL.draw_photometric_params(4, 224, {"hflip": {"use": True}}, random.Random(0)) == []
    with pytest.raises(NotImplementedError):
        DL.draw_augment_params(2, (256, 256), 224, True, {"rotate": {"use": True}, "hflip": {"use": True}})
