"""Pipeline-level tests (-m gpu): dataset side (normalise/crop/tiling), run.py modes, chip + sliding-window inference."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from instageo_amd import dataloader as DL  # noqa: E402
from instageo_amd import ops  # noqa: E402
from instageo_amd.infer_utils import chip_inference, sliding_window_inference, stitch_windows  # noqa: E402
from instageo_amd.model import PrithviSeg  # noqa: E402
from oracle import prithvi_oracle as O  # noqa: E402

DEV = "cuda"
MEAN = [0.14245495, 0.13921481, 0.12434631, 0.31420089, 0.20743526, 0.12046503]
STD = [0.04036231, 0.04186983, 0.05267646, 0.0822221, 0.06834774, 0.05294205]


def test_process_test_shapes_and_values():
    """Reference tests/model_tests/test_dataloader.py:151-160: 512^2 chip -> (4, C, 1, 224, 224) windows."""
    rng = np.random.default_rng(0)
    x = rng.integers(0, 10000, size=(6, 512, 512)).astype(np.int16)
    y = rng.integers(0, 2, size=(512, 512)).astype(np.float32)
    imgs, labels = DL.process_test(x, y, MEAN, STD, temporal_size=1, img_size=512, crop_size=224, stride=224, constant_multiplier=1e-4)
    assert imgs.shape == (4, 6, 1, 224, 224) and labels.shape == (4, 224, 224)
    wins = O.window_origins(512, 224, 224)
    for i, (t, l) in enumerate(wins):
        ref = O.normalize_chip(x[:, t : t + 224, l : l + 224].astype(np.float64) * 1e-4, MEAN, STD, 1)
        assert np.allclose(imgs[i].cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
        assert np.array_equal(labels[i].cpu().numpy(), y[t : t + 224, l : l + 224])
    # crop_array exact values (test_dataloader.py:117-148)
    a = np.arange(24).reshape(2, 3, 4)
    assert np.array_equal(DL.crop_array(a, 1, 0, 3, 2), a[:, 0:2, 1:3])
    with pytest.raises(ValueError):
        DL.crop_array(np.zeros((1, 1, 1, 1, 1)), 0, 0, 1, 1)


def test_process_and_augment_crop_flip_layout():
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 10000, (18, 256, 256), dtype=torch.int16)
    y = torch.randint(0, 13, (256, 256)).float()
    out, lab = DL.process_and_augment(x, y, [500.0] * 6, [300.0] * 6, temporal_size=3, im_size=224,
                                      augmentations={"hflip": {"use": True, "p": 1.0}, "vflip": {"use": False}}, generator=g)
    assert out.shape == (6, 3, 224, 224) and lab.shape == (224, 224) and out.dtype == torch.float32
    # undo the flip and find the crop: band index t*6+c maps to out[c, t]
    o = out.flip(-1).cpu()
    xs = (x.float() - 500.0) / 300.0
    g2 = torch.Generator().manual_seed(3)
    top = int(torch.randint(0, 33, (1,), generator=g2))
    left = int(torch.randint(0, 33, (1,), generator=g2))
    assert torch.allclose(o[2, 1], xs[1 * 6 + 2, top : top + 224, left : left + 224], atol=1e-5)
    with pytest.raises(NotImplementedError):
        DL.process_and_augment(x, y, [0.0] * 6, [1.0] * 6, 3, augmentations={"rotate": {"use": True}})


def _tiny(ncls=2, T=1):
    net = PrithviSeg(temporal_step=T, num_classes=ncls, load_pretrained_weights=False, freeze_backbone=True, variant="prithvi_eo_tiny", device=DEV)
    cfg = O.make_config("prithvi_eo_tiny", T, ncls)
    sd = O.make_state_dict(cfg, seed=11)
    net.load_state_dict(sd)
    return net, cfg, sd


def test_chip_inference_writes_int8_maps(tmp_path):
    net, cfg, sd = _tiny()
    ds = DL.SyntheticChipDataset(5, 1, 2, MEAN, STD, device=DEV)
    arr = DL.ArrayChipDataset([ds.raw(i)[0] for i in range(5)], [ds.raw(i)[1] for i in range(5)], MEAN, STD, 1, 1e-4,
                              include_filenames=True, names=[f"chip_{i}.tif" for i in range(5)], device=DEV)
    loader = [DL.infer_collate_fn([arr[i] for i in range(s, min(s + 2, 5))]) for s in range(0, 5, 2)]
    assert chip_inference(loader, str(tmp_path), net, device="gpu") == {}
    files = sorted(os.listdir(tmp_path))
    assert files == [f"prediction_chip_{i}.npy" for i in range(5)]
    pred = np.load(tmp_path / files[3])
    assert pred.dtype == np.int8 and pred.shape == (224, 224)
    with torch.no_grad():
        x, _ = arr[3][0]
        ref = O.prithvi_seg_forward(cfg, sd, x.cpu()[None], training=False).argmax(1)[0].numpy()
    assert (pred == ref).mean() > 0.99  # bf16 near-ties may flip a few pixels


def test_sliding_window_inference_matches_per_window_forward():
    """configs[3] semantics on a small tile: S=700 -> 3x3 windows of 224, 28-px remainder dropped."""
    net, cfg, sd = _tiny()
    g = torch.Generator(device=DEV).manual_seed(5)
    tile = torch.randint(0, 10000, (6, 700, 700), generator=g, device=DEV, dtype=torch.int16)
    maps, origins = sliding_window_inference(tile, net, MEAN, STD, 1, 224, 224, batch_size=4, constant_multiplier=1e-4)
    assert len(origins) == 9 and maps.shape == (9, 224, 224) and maps.dtype == torch.int8
    t, l = origins[5]
    x = DL.normalize_batch(tile[None, :, t : t + 224, l : l + 224].contiguous(), MEAN, STD, 1, 1e-4)
    with torch.no_grad():
        direct = ops.argmax_i8(net(x))[0]
    assert torch.equal(maps[5], direct)
    canvas = stitch_windows(maps, origins, 700)
    assert canvas.shape == (700, 700) and torch.equal(canvas[t : t + 224, l : l + 224], maps[5]) and int(canvas[699, 699]) == -1
    # size-independent property at the full BASELINE size: window list of a 10980^2 tile
    assert len(DL.window_origins(10980, 224, 224)) == 2401


def test_run_train_eval_chip_inference(tmp_path, capsys):
    from instageo_amd import run

    common = ["model.model_name=prithvi_eo_tiny", "model.load_pretrained_weights=False", "train.batch_size=2", "train.ignore_index=-1",
              "train.class_weights=[1,3]", f"root_dir={tmp_path}"]
    out = str(tmp_path / "out")
    rc = run.main(["--output-dir", out, "mode=train", "train.num_epochs=2", "train_filepath=synthetic:6", "valid_filepath=synthetic:4"] + common)
    assert rc == 0
    ck = os.path.join(out, "instageo_best_checkpoint.ckpt")
    assert os.path.exists(ck) and os.path.exists(os.path.join(out, ".hydra", "config.yaml"))
    sd = torch.load(ck)["state_dict"]
    assert "net.prithvi_encoder.pos_embed" in sd and "criterion.weight" in sd
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 2 and {"train_loss", "val_loss", "train_IoU", "val_IoU", "train_Acc", "val_F1", "learning_rate"} <= set(lines[0])
    rc = run.main(["--output-dir", out, "mode=eval", "test_filepath=synthetic:2", "test.img_size=448", f"checkpoint_path={ck}"] + common)
    assert rc == 0
    res = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")][-1]["Evaluation results"]
    assert {"test_loss", "test_IoU", "test_Acc"} <= set(res) and 0 <= res["test_IoU"] <= 1
    rc = run.main(["--output-dir", out, "mode=chip_inference", "test_filepath=synthetic:3", "test.img_size=224", f"checkpoint_path={ck}"] + common)
    assert rc == 0 and len(os.listdir(tmp_path / "predictions")) == 3
    with pytest.raises(RuntimeError):
        run.main(["--output-dir", out, "mode=eval", "test_filepath=synthetic:2"] + common)  # checkpoint_path required
