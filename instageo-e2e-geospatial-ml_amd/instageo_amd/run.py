"""Runner entry point with the reference's ``run.py`` surface (``instageo/model/run.py:60-245``).

    python -m instageo_amd.run [--config-name sen1floods11] [--config-path DIR] key=value ...

Modes ``stats | train | eval | chip_inference`` and every config key are those of the reference.  Hydra,
Lightning and Neptune are replaced by :mod:`instageo_amd.config` and the explicit loop below, which logs the
same metric names and writes ``instageo_best_checkpoint.ckpt`` (``{"state_dict": ...}``) on the best
``val_IoU`` (pipeline_utils.py:347-355).  Data: ``*_filepath`` may be ``synthetic:<n>`` (on-device HLS-shaped
chips), an ``.npz`` with ``chips (N,T*C,H,W)`` and ``labels (N,H,W)``, or the reference's own CSV of chip / label GeoTIFF paths
(``InstaGeoDataset`` on the host TIFF codec), relative to ``root_dir``.  Multi-GPU: launch with ``python -m torch.distributed.run``; ranks shard the
dataset and average gradients over RCCL (:mod:`instageo_amd.distributed`).
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import distributed as D
from .config import check_required_flags, load_config
from .dataloader import (ArrayChipDataset, InstaGeoDataset, SyntheticChipDataset, eval_collate_fn, infer_collate_fn, normalize_batch,
                         process_and_augment, process_and_augment_batch, process_test)
from .factory import create_model
from .infer_utils import chip_inference
from .pipeline_utils import compute_stats

SEED = 1042  # pl.seed_everything(1042) in the reference (run.py:50)


def get_device() -> str:
    """pipeline_utils.py:58-75 returns gpu/cpu; there is no CPU path here."""
    if not torch.cuda.is_available():
        raise RuntimeError("instageo_amd needs a HIP device (MI355X): no CPU fallback")
    return "gpu"


def create_dataset(spec: Optional[str], cfg: Dict[str, Any], kind: str, device: str):
    """``synthetic:<n>`` or an .npz archive -> dataset following the reference item contract."""
    d = cfg["dataloader"]
    T, mean, std = d["temporal_dim"], d["mean"], d["std"]
    mult = d.get("constant_multiplier", 1.0)
    mult = None if mult in (None, 1, 1.0) else float(mult)
    ncls, ign = cfg["model"]["num_classes"], cfg["train"]["ignore_index"]
    if spec is None or str(spec) == "None":
        raise RuntimeError(f"{kind}_filepath is required")
    if str(spec).startswith("synthetic:"):
        n = int(str(spec).split(":")[1])
        size = cfg["test"]["img_size"] if kind == "test" and cfg["mode"] == "eval" else d["img_size"]
        return SyntheticChipDataset(n, T, ncls, mean, std, im_size=size, ignore_index=ign, constant_multiplier=1e-4, seed=SEED, device=device,
                                    regression=bool(cfg.get("is_reg_task", False)))
    path = spec if os.path.isabs(str(spec)) or cfg.get("root_dir") in (None, "None") else os.path.join(cfg["root_dir"], spec)
    if str(path).endswith(".csv"):
        # the reference's own input: a CSV of chip / label GeoTIFF paths (dataloader.py:786-906) through the TIFF codec
        from functools import partial

        t = cfg["test"]
        if kind == "test" and cfg["mode"] == "eval":
            pre = partial(process_test, mean=mean, std=std, temporal_size=T, img_size=t["img_size"], crop_size=t["crop_size"],
                          stride=t["stride"], device=device)
        else:
            size = t["img_size"] if kind == "test" else d["img_size"]
            pre = partial(process_and_augment, mean=mean, std=std, temporal_size=T, im_size=size, crop=(kind != "test"),
                          augmentations=None, device=device)
        return InstaGeoDataset(path, cfg.get("root_dir") or "", pre, d.get("no_data_value", -9999), ign, d.get("replace_label"),
                               bool(d.get("reduce_to_zero", False)), 1.0 if mult is None else mult, d.get("bands"),
                               include_filenames=(kind == "test"), mean=mean, std=std, temporal_size=T, device=device)
    z = np.load(path)
    return ArrayChipDataset(z["chips"], z["labels"], mean, std, T, mult, include_filenames=(kind == "test"), device=device,
                            replace_label=d.get("replace_label"), reduce_to_zero=bool(d.get("reduce_to_zero", False)))


def shard_indices(n: int, shuffle: bool, epoch: int, rank: int, world: int, equal: bool) -> List[int]:
    """Item indices of one rank for one epoch.

    ``equal=True`` is ``torch.utils.data.DistributedSampler`` (what Lightning DDP gives the reference's train loader): the
    (shuffled) index list is padded by wrap-around to ``ceil(n / world) * world`` and rank r takes ``idx[r::world]``, so EVERY
    rank gets the same number of items and therefore the same number of optimizer steps -- a rank with one batch fewer would
    leave the others waiting in a gradient all-reduce while it has already entered the epoch-end metric all-reduce.
    ``equal=False`` is an exact contiguous partition (no duplicates; sizes differ by at most one): used for validation /
    test, whose steps contain no collective, so that the reduced metrics count every item exactly once."""
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(SEED + epoch)).tolist() if shuffle else list(range(n))
    if not equal or world == 1:
        lo, hi = D.shard_range(n, rank, world)
        return idx[lo:hi]
    if n == 0:
        return []
    total = -(-n // world) * world
    while len(idx) < total:
        idx += idx[: total - len(idx)]
    return idx[rank:total:world]


def _batches(ds, batch_size: int, shuffle: bool, epoch: int, rank: int, world: int, equal: bool = False):
    """Rank-sharded batch index lists (same shuffle seed on all ranks; see :func:`shard_indices`)."""
    idx = shard_indices(len(ds), shuffle, epoch, rank, world, equal)
    for i in range(0, len(idx), batch_size):
        yield idx[i : i + batch_size]


def _stack(ds, ids: List[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    items = [ds[i] for i in ids]
    items = [it[0] if isinstance(it[0], tuple) else it for it in items]
    return torch.stack([it[0] for it in items]), torch.stack([it[1] for it in items])


def _reduce_metrics(metrics, dev, model=None, step_type: Optional[str] = None) -> None:
    """Sum the rank-local streaming state over ranks before the epoch-end hooks: the confusion matrix (or the regression
    sums), the (sum of batch losses, #batches) pair behind ``<step>_loss`` and, for the test epoch, the ROC-AUC histograms."""
    D.reduce_confusion(metrics.device_matrix(dev) if hasattr(metrics, "device_matrix") else metrics.device_sums(dev))
    if model is None or not D.dp_active():
        return
    acc = model._loss_sums.get(step_type)
    if acc is None:  # a rank whose shard was empty still has to take part in the collective
        acc = torch.zeros(2, dtype=torch.float64, device=dev)
        model._loss_sums[step_type] = acc
    D.reduce_loss_stats(acc)
    if step_type == "test" and hasattr(model, "test_auc"):
        D.reduce_confusion(model.test_auc.device_hist(dev))


def train(cfg: Dict[str, Any], model, out_dir: str, rank: int, world: int) -> Dict[str, float]:
    dev = str(model.net.store.flat.device)
    train_ds = create_dataset(cfg["train_filepath"], cfg, "train", dev)
    valid_ds = create_dataset(cfg["valid_filepath"], cfg, "valid", dev)
    bs = cfg["train"]["batch_size"]
    opt = model.optimizer()
    sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2, eta_min=0) if cfg["train"].get("scheduler") else None
    D.attach_data_parallel(model)
    reg = bool(cfg.get("is_reg_task", False))
    best, history = (-float("inf") if reg else -1.0), {}
    augs = cfg["dataloader"].get("augmentations") or {}
    train_augs = {k: v for k, v in augs.items() if v.get("use", False)}  # config order = order of application
    aug_rng = random.Random(SEED + 104729 * rank)  # rotate / brightness / blur / noise draws (Python's random, like the reference)
    aug_gen = torch.Generator().manual_seed(SEED + 7919 * rank)  # different crops/flips per rank, reproducible
    for epoch in range(cfg["train"]["num_epochs"]):
        model.net.train()
        for ids in _batches(train_ds, bs, True, epoch, rank, world, equal=True):
            # training items go through process_and_augment (dataloader.py:527-585): random crop to img_size + the enabled
            # flips + normalise as ONE kernel per batch on the raw chips; rotate / brightness / blur / noise (off in
            # sen1floods11.yaml) add one ig_aug_* launch each between the crop and the normalisation
            xr, yr = train_ds.raw_batch(ids)
            x, y = process_and_augment_batch(xr, yr, train_ds.mean, train_ds.std, train_ds.T, cfg["dataloader"]["img_size"], True,
                                             train_augs, train_ds.mult, aug_gen,
                                             label_no_data_value=cfg["train"].get("ignore_index", -1),  # run.py:128-129
                                             chip_no_data_value=cfg["dataloader"].get("no_data_value", -9999),
                                             max_pixel_value=cfg["dataloader"].get("max_pixel_value", 10000.0), rng=aug_rng)
            model.fused_train_step(x, y)
        _reduce_metrics(model.train_metrics, dev, model, "train")
        model.on_train_epoch_end()
        for ids in _batches(valid_ds, bs, False, epoch, rank, world):
            x, y = _stack(valid_ds, ids)
            model.fused_eval_step(x, y, "val")
        _reduce_metrics(model.val_metrics, dev, model, "val")
        model.on_validation_epoch_end()
        if sched is not None:
            sched.step()
        model.log("learning_rate", opt.param_groups[0]["lr"])
        history = {k: (float(v) if not isinstance(v, (list, tuple)) else v) for k, v in model.logged.items()}
        model.sync_master_params()  # data parallel: the fp32 masters are sharded between checkpoints (collective, every rank)
        if rank == 0:
            print(json.dumps({"epoch": epoch, **{k: round(v, 6) for k, v in history.items() if isinstance(v, float)}}))
            # ModelCheckpoint(monitor = "val_RMSE" (min) for regression else "val_IoU" (max), save_top_k=1)  (run.py:163-164)
            score = -history.get("val_RMSE", float("inf")) if reg else history.get("val_IoU", 0.0)
            if score > best:
                best = score
                torch.save({"state_dict": model.checkpoint_state_dict(), "epoch": epoch}, os.path.join(out_dir, "instageo_best_checkpoint.ckpt"))
    return history


def evaluate(cfg: Dict[str, Any], model, rank: int, world: int) -> Dict[str, float]:
    dev = str(model.net.store.flat.device)
    test_ds = create_dataset(cfg["test_filepath"], cfg, "test", dev)
    d, t = cfg["dataloader"], cfg["test"]
    model.net.eval()
    lo, hi = D.shard_range(len(test_ds), rank, world)
    for i in range(lo, hi):
        raw_x, raw_y = test_ds.raw(i) if hasattr(test_ds, "raw") else (test_ds.chips[i], test_ds.labels[i])
        mult = getattr(test_ds, "mult", None)  # the constant multiplier applies in every mode (dataloader.py:707-750)
        x, y = process_test(raw_x, raw_y, d["mean"], d["std"], d["temporal_dim"], t["img_size"], t["crop_size"], t["stride"], mult, dev)
        model.fused_eval_step(x, y, "test")
    _reduce_metrics(model.test_metrics, dev, model, "test")
    model.on_test_epoch_end()
    return {k: float(v) for k, v in model.logged.items() if k.startswith("test_") and not isinstance(v, (list, tuple))}


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config-name", default="config")
    ap.add_argument("--config-path", default=None)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3"])
    ap.add_argument("--output-dir", default=None, help="replaces the Hydra run dir (checkpoint + resolved config)")
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args(argv)
    cfg = load_config(args.config_name, args.overrides, args.config_path)
    start = time.time()
    torch.manual_seed(SEED)
    np.random.seed(SEED)
    rank, local_rank, world = D.init_from_env()
    get_device()
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    out_dir = args.output_dir or os.path.join(os.getcwd(), "outputs", time.strftime("%Y-%m-%d_%H-%M-%S"))
    if rank == 0:
        os.makedirs(os.path.join(out_dir, ".hydra"), exist_ok=True)
        import yaml

        yaml.safe_dump(cfg, open(os.path.join(out_dir, ".hydra", "config.yaml"), "w"), sort_keys=False)

    if cfg["mode"] == "stats":
        # run.py:89-111: the train set with mean 0 / std 1 and no augmentation, reduced to per-band mean/std + class weights
        check_required_flags(["train_filepath"], cfg)
        scfg = {**cfg, "dataloader": {**cfg["dataloader"], "mean": [0.0] * len(cfg["dataloader"]["mean"]),
                                      "std": [1.0] * len(cfg["dataloader"]["std"])}}  # fmt: skip
        ds = create_dataset(cfg["train_filepath"], scfg, "train", dev)
        bs = cfg["train"]["batch_size"]
        loader = (_stack(ds, ids) for ids in _batches(ds, bs, False, 0, 0, 1))
        mean, std, class_weights = compute_stats(loader, is_reg_task=bool(cfg.get("is_reg_task", False)), device=dev)
        if rank == 0:
            print(json.dumps({"mean": mean, "std": std, "class_weights": class_weights}))
        return 0
    model = create_model(cfg, precision=args.precision, device=dev)
    if cfg["mode"] == "train":
        check_required_flags(["train_filepath", "valid_filepath"], cfg)
        hist = train(cfg, model, out_dir, rank, world)
        if rank == 0:
            print(f"Elapsed time: {time.time() - start:.2f} seconds")
    elif cfg["mode"] == "eval":
        check_required_flags(["test_filepath", "checkpoint_path"], cfg)
        res = evaluate(cfg, model, rank, world)
        if rank == 0:
            print(json.dumps({"Evaluation results": {k: round(v, 6) for k, v in res.items()}}))
            print(f"Elapsed time: {time.time() - start:.2f} seconds")
    elif cfg["mode"] == "chip_inference":
        check_required_flags(["root_dir", "test_filepath", "checkpoint_path"], cfg)
        model.net.eval()
        output_dir = os.path.join(cfg["root_dir"], "predictions")
        ds = create_dataset(cfg["test_filepath"], cfg, "test", dev)
        if isinstance(ds, SyntheticChipDataset):
            ds = ArrayChipDataset([ds.raw(i)[0] for i in range(len(ds))], [ds.raw(i)[1] for i in range(len(ds))], ds.mean, ds.std, ds.T,
                                  1e-4, include_filenames=True, device=dev)
        lo, hi = D.shard_range(len(ds), rank, world)
        bs = cfg["train"]["batch_size"]
        loader = (infer_collate_fn([ds[j] for j in range(i, min(i + bs, hi))]) for i in range(lo, hi, bs))
        info = chip_inference(loader, output_dir, model, device="gpu")
        if rank == 0:
            print(f"Carbon tracking information: {info}")
    else:
        raise ValueError(f"unknown mode {cfg['mode']!r}")
    if D.dp_active():
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
