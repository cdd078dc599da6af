"""Host-side API the reference's own tests exercise next to the hot path (tests/model_tests/test_utils.py, test_dataloader.py,
test_pipeline_utils.py): checkpoint plumbing, CSV / GeoTIFF chip loading, loader helpers.  No GPU needed."""
import os

import numpy as np
import pytest
import torch

from instageo_amd import dataloader as DL
from instageo_amd import pipeline_utils as P
from instageo_amd import tiff
from instageo_amd import utils as U

TIFS = os.path.join(os.path.dirname(__file__), "golden", "tiff")
CHIP, MASK = os.path.join(TIFS, "chip_178_022.tif"), os.path.join(TIFS, "chip_178_022.mask.tif")


# ---- utils.py: checkpoint plumbing --------------------------------------------------------------------------------------
def test_get_state_dict_picks_the_first_state_dict_key_or_returns_the_dict():
    inner = {"layer1.weight": torch.zeros(2, 2)}
    assert U.get_state_dict({"model.state_dict": inner, "other_key": "value"}) is inner
    plain = {"layer1.weight": torch.zeros(2, 2), "layer1.bias": torch.zeros(2)}
    assert U.get_state_dict(plain) is plain
    first = {"a": torch.zeros(1)}
    assert U.get_state_dict({"first.state_dict": first, "second.state_dict": {"b": torch.zeros(1)}}) is first


def test_common_prefix_proj_key_and_prefix_removal():
    assert U.get_common_prefix(["model.layer1.weight", "model.layer2.bias", "model.layer3.activation_func"]) == "model."
    sd = {"model.patch_embed.proj.weight": torch.zeros(2, 2), "model.patch_embed.proj.bias": torch.zeros(2)}
    assert U.get_proj_key(sd, return_prefix=True) == ("model.patch_embed.proj.weight", "model.")
    assert U.get_proj_key(sd) == ("model.patch_embed.proj.weight", None)
    sd2 = {"model.patch_embed.projection.weight": torch.zeros(2, 2)}
    assert U.get_proj_key(sd2, return_prefix=True) == ("model.patch_embed.projection.weight", "model.")
    assert U.get_proj_key({"model.layer1.weight": torch.zeros(1)}) == (None, None)
    stripped = U.remove_prefixes({"model.layer1.weight": 1, "model.layer1.bias": 2}, "model.")
    assert set(stripped) == {"layer1.weight", "layer1.bias"}


def test_patch_embed_compatibility_ignores_only_the_band_axis():
    f = U.patch_embed_weights_are_compatible
    assert f(torch.zeros(64, 6, 16, 16), torch.zeros(64, 3, 16, 16))
    assert not f(torch.zeros(64, 6, 16, 16), torch.zeros(64, 3, 32, 32))
    assert not f(torch.zeros(32, 6, 16, 16), torch.zeros(64, 3, 16, 16))
    assert not f(torch.zeros(64, 6, 16, 16), torch.zeros(64, 3, 16))


class _MockViT(torch.nn.Module):
    def __init__(self, temporal_encoding=True, location_encoding=True):
        super().__init__()
        self.temporal_encoding, self.location_encoding = temporal_encoding, location_encoding
        self.pos_embed = torch.randn(1, 196, 768)
        self.patch_embed = torch.nn.Module()
        self.patch_embed.proj = torch.nn.Conv2d(6, 768, kernel_size=16, stride=16)


@pytest.mark.parametrize("coords", [True, False])
def test_checkpoint_filter_fn_vit(coords):
    model = _MockViT(coords, coords)
    ck_w = torch.randn(768, 3, 16, 16)
    sd = {"patch_embed.proj.weight": ck_w, "pos_embed": torch.randn(1, 196, 768), "temporal_embed": torch.randn(1, 12, 768),
          "location_embed": torch.randn(1, 2, 768), "decoder.blocks.0.weight": torch.randn(4, 4), "encoder.blocks.0.weight": torch.randn(4, 4),
          "_timm_module.norm.weight": torch.randn(4), "decoder_pos_embed": torch.randn(1, 4, 4), "mask_token": torch.randn(1, 1, 4)}
    pretrained, bands = ["RED", "GREEN", "BLUE"], ["RED", "GREEN", "BLUE", "NIR_NARROW", "SWIR_1", "SWIR_2"]
    out = U.checkpoint_filter_fn_vit(dict(sd), model, pretrained, bands)
    assert "decoder.blocks.0.weight" not in out and "mask_token" not in out and "decoder_pos_embed" not in out
    assert "blocks.0.weight" in out and "encoder.blocks.0.weight" not in out and "norm.weight" in out
    assert out["pos_embed"] is model.pos_embed  # the model's own fixed table, not the checkpoint's
    assert ("temporal_embed" in out) == coords and ("location_embed" in out) == coords
    w = out["patch_embed.proj.weight"]
    assert w.shape == (768, 6, 16, 16)  # the model's band count
    for i in range(3):  # pretrained bands copied in place, the others keep a fresh initialisation
        assert torch.equal(w[:, i], ck_w[:, i])
    assert float(w[:, 3:].abs().sum()) > 0 and not torch.equal(w[:, 3], ck_w[:, 0])


def test_select_patch_embed_weights_reorders_and_strips_a_prefix():
    model = _MockViT()
    ck_w = torch.randn(768, 6, 16, 16)
    out = U.select_patch_embed_weights({"backbone.patch_embed.proj.weight": ck_w, "backbone.norm.weight": torch.ones(2)}, model,
                                       [0, 1, 2, 3, 4, 5], [5, 4, 3, 2, 1, 0])
    assert set(out) == {"patch_embed.proj.weight", "norm.weight"}
    for i in range(6):
        assert torch.equal(out["patch_embed.proj.weight"][:, i], ck_w[:, 5 - i])
    with pytest.raises(Exception, match="patch embed"):
        U.select_patch_embed_weights({"norm.weight": torch.ones(2)}, model, [0], [0])
    # incompatible patch size: the model's weight is kept
    kept = U.select_patch_embed_weights({"patch_embed.proj.weight": torch.randn(768, 6, 8, 8)}, model, [0, 1, 2, 3, 4, 5], [0, 1, 2, 3, 4, 5])
    assert torch.equal(kept["patch_embed.proj.weight"], model.state_dict()["patch_embed.proj.weight"])


# ---- dataloader.py: rasters, CSV lists, dataset ----------------------------------------------------------------------------
def test_get_raster_data_and_process_data_on_the_reference_test_chip():
    x = DL.get_raster_data(CHIP, is_label=False)
    assert x.shape == (18, 32, 32) and x.dtype == np.int16
    sel = DL.get_raster_data(CHIP, is_label=False, bands=[0, 6, 12])
    assert sel.shape == (3, 32, 32) and np.array_equal(sel, x[[0, 6, 12]])
    y = DL.get_raster_data(MASK)
    assert y.shape == (1, 32, 32)
    assert np.array_equal(DL.get_raster_data(MASK, is_label=True, bands=[0]), y)  # band selection never applies to labels
    with pytest.raises(NotImplementedError):
        DL.get_raster_data({"tiles": {}})
    ax, ay = DL.process_data(CHIP, MASK, bands=[0, 1, 2], constant_multiplier=1e-4, replace_label=(1, 9), reduce_to_zero=True)
    assert ax.shape == (3, 32, 32) and np.allclose(ax, x[:3] * 1e-4)
    exp = np.where(y == 1, 9, y) - 1
    assert np.array_equal(ay, exp) and (y == 1).any()
    ax2, ay2 = DL.process_data(CHIP)
    assert ay2 is None and np.array_equal(ax2, x * 1.0)


def _write_set(tmp_path):
    chip = np.full((6, 16, 16), 500, np.int16)
    chip[:, :8] = -9999  # upper half: chip NODATA
    lab_ok = np.full((16, 16), -1, np.int16)
    lab_ok[12, 3] = 1  # one valid pixel over chip data
    lab_masked = np.full((16, 16), -1, np.int16)
    lab_masked[2, 3] = 1  # valid label only where the chip has no data
    for name, arr in (("chip_a.tif", chip), ("chip_b.tif", chip), ("lab_a.tif", lab_ok[None]), ("lab_b.tif", lab_masked[None])):
        tiff.write(str(tmp_path / name), arr)
    csv = tmp_path / "set.csv"
    csv.write_text("Input,Label\nchip_a.tif,lab_a.tif\nchip_b.tif,lab_b.tif\nmissing.tif,lab_a.tif\n")
    return csv


def test_mask_label_with_chip_and_get_valid_filepaths(tmp_path, capsys):
    csv = _write_set(tmp_path)
    root = str(tmp_path)
    assert DL.mask_label_with_chip(os.path.join(root, "chip_a.tif"), os.path.join(root, "lab_a.tif"), -9999, -1) is False
    assert DL.mask_label_with_chip(os.path.join(root, "chip_b.tif"), os.path.join(root, "lab_b.tif"), -9999, -1) is True
    paths = DL.get_valid_filepaths(str(csv), root, no_data_value=-9999, ignore_index=-1)
    assert paths == [(os.path.join(root, "chip_a.tif"), os.path.join(root, "lab_a.tif"))]
    assert "Dropped a total of 2 rows" in capsys.readouterr().out
    only_inputs = tmp_path / "inputs.csv"
    only_inputs.write_text("Input\nchip_a.tif\nchip_b.tif\n")
    assert DL.get_valid_filepaths(str(only_inputs), root) == [(os.path.join(root, "chip_a.tif"), None), (os.path.join(root, "chip_b.tif"), None)]


def test_instageo_dataset_items_and_collate(tmp_path):
    csv = _write_set(tmp_path)
    pre = lambda x, y: (torch.as_tensor(x, dtype=torch.float32).unsqueeze(1), None if y is None else torch.as_tensor(y[0], dtype=torch.float32))
    ds = DL.InstaGeoDataset(str(csv), str(tmp_path), pre, chip_no_data_value=-9999, label_no_data_value=-1, replace_label=(-1, -1),
                            reduce_to_zero=False, constant_multiplier=0.5, include_filenames=True)
    assert len(ds) == 1
    (x, y), fname, nodata = ds[0]
    assert x.shape == (6, 1, 16, 16) and y.shape == (16, 16) and fname.endswith("chip_a.tif")
    assert float(x[0, 0, 12, 3]) == 250.0  # constant multiplier applied before the preprocess function
    assert nodata.shape == (6, 16, 16) and not nodata.any()  # 0.5 * -9999 != -9999: the mask is taken AFTER the multiplier, like the reference
    (data, labels), files = P.infer_collate_fn([ds[0], ds[0]])
    assert data.shape == (2, 6, 1, 16, 16) and len(labels) == 2 and files == [fname, fname]
    plain = DL.InstaGeoDataset(str(csv), str(tmp_path), pre, -9999, -1, None, False, 1.0)
    xs, ys = P.eval_collate_fn([(plain[0],), (plain[0],)])
    assert xs.shape == (12, 1, 16, 16) and ys.shape == (32, 16)  # torch.cat over dim 0, as the reference does for test windows
    loader = P.create_dataloader(plain, batch_size=1, shuffle=False, num_workers=0, pin_memory=False)
    assert len(loader) == 1 and next(iter(loader))[0].shape == (1, 6, 1, 16, 16)


# ---- pipeline_utils.py ----------------------------------------------------------------------------------------------------
def test_check_required_flags_and_device():
    P.check_required_flags(["root_dir"], {"root_dir": "/data"})
    with pytest.raises(RuntimeError, match="--root_dir is required"):
        P.check_required_flags(["root_dir"], {"root_dir": "None"})

    class Cfg:
        valid_filepath = "None"

    with pytest.raises(RuntimeError, match="--valid_filepath"):
        P.check_required_flags(["valid_filepath"], Cfg())
    assert P.get_device() in ("gpu", "cpu")
    assert P.compute_class_weights({0: 10, 1: 30}) == [2.0, 2.0 / 3.0]


# ---- model.py: position tables (tests/model_tests/test_model.py scenarios) ------------------------------------------------
def test_sincos_tables_and_interpolation():
    from instageo_amd.model import (get_1d_sincos_embed_from_grid_torch, get_1d_sincos_pos_embed_from_grid, get_3d_sincos_pos_embed,
                                    interpolate_pos_encoding)

    e = get_1d_sincos_pos_embed_from_grid(8, np.array([0, 1, 2, 3]))
    assert e.shape == (4, 8) and np.allclose(e[0], [0, 0, 0, 0, 1, 1, 1, 1])
    assert get_1d_sincos_pos_embed_from_grid(8, np.array([0.5, 1.5])).shape == (2, 8)
    et = get_1d_sincos_embed_from_grid_torch(8, torch.tensor([0.0, 1.0, 2.0, 3.0]))
    assert et.shape == (4, 8) and np.allclose(et.numpy(), e, atol=1e-6)
    grid = (2, 3, 4)
    pe = get_3d_sincos_pos_embed(32, grid, cls_token=False)
    assert pe.shape == (24, 32)
    pc = get_3d_sincos_pos_embed(32, grid, cls_token=True)
    assert pc.shape == (25, 32) and np.allclose(pc[0], 0) and np.allclose(pc[1:], pe)
    table = torch.randn(1, 25, 32)
    assert interpolate_pos_encoding(table, grid, (1, 1, 1), (2, 3, 4), 32) is table  # same token grid: untouched
    up = interpolate_pos_encoding(table, grid, (1, 1, 1), (2, 6, 8), 32)
    assert up.shape == (1, 1 + 2 * 6 * 8, 32) and torch.equal(up[:, :1], table[:, :1])
    # align_corners=True: the corner tokens of every frame are kept exactly
    body, ub = table[0, 1:].reshape(2, 3, 4, 32), up[0, 1:].reshape(2, 6, 8, 32)
    assert torch.allclose(ub[:, 0, 0], body[:, 0, 0], atol=1e-6) and torch.allclose(ub[:, -1, -1], body[:, -1, -1], atol=1e-6)
    # another frame count: the table is regenerated for the new number of frames before the spatial resampling
    more = interpolate_pos_encoding(table, grid, (1, 1, 1), (4, 3, 4), 32)
    assert more.shape == (1, 49, 32)
    assert np.allclose(more[0].numpy(), get_3d_sincos_pos_embed(32, (4, 3, 4), cls_token=True), atol=1e-6)
