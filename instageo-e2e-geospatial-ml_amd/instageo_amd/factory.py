"""Model factory (reference: ``instageo/model/factory.py:35-116``): config -> task module (+ checkpoint)."""
from __future__ import annotations

from typing import Any, Dict

import torch

from .regression import PrithviDistillationRegressionModule, PrithviRegressionModule
from .segmentation import PrithviDistillationSegmentationModule, PrithviSegmentationModule


def create_model(cfg: Dict[str, Any], precision: str = "bf16", device=None) -> PrithviSegmentationModule:
    """Build the segmentation (or, with ``is_reg_task``, regression) module from a run.py config; non-train modes load ``checkpoint_path`` strictly
    (``torch.load(path)["state_dict"]``, factory.py:113-115)."""
    m, t, d = cfg["model"], cfg["train"], cfg["dataloader"]
    train_mode = cfg["mode"] == "train"
    common = dict(
        image_size=d["img_size"] if train_mode else cfg["test"]["crop_size"],
        learning_rate=t["learning_rate"],
        freeze_backbone=m["freeze_backbone"],
        load_pretrained_weights=bool(m["load_pretrained_weights"]) and train_mode and cfg.get("allow_hub_download", False),
        temporal_step=d["temporal_dim"],
        ignore_index=t["ignore_index"],
        weight_decay=t["weight_decay"],
        scheduler=t.get("scheduler", False),
        model_name=m["model_name"],
        weight_clip_range=m.get("weight_clip_range"),
        depth=m.get("depth", -1),
        precision=precision,
        device=device,
    )
    distill = bool(t.get("distillation", False)) and train_mode
    if distill and cfg.get("is_reg_task", False):  # factory.py:61-69
        common_d = {k: v for k, v in common.items() if k != "depth"}
        model = PrithviDistillationRegressionModule(teacher_ckpt_path=t["teacher_ckpt_path"], depth=t.get("teacher_depth", -1),
                                                    student_depth=m.get("depth", -1), use_log_scale=m.get("use_log_scale", False),
                                                    plot_reg_results=m.get("plot_reg_results", False),
                                                    include_ee=m.get("include_ee_metric", False), **common_d)
    elif distill:  # factory.py:84-91: frozen teacher from train.teacher_ckpt_path, student of model.depth blocks
        common_d = {k: v for k, v in common.items() if k != "depth"}
        model = PrithviDistillationSegmentationModule(teacher_ckpt_path=t["teacher_ckpt_path"], num_classes=m["num_classes"],
                                                      class_weights=t["class_weights"], depth=t.get("teacher_depth", -1),
                                                      student_depth=m.get("depth", -1), **common_d)
    elif cfg.get("is_reg_task", False):  # factory.py:58-76, 97-104
        model = PrithviRegressionModule(use_log_scale=m.get("use_log_scale", False), plot_reg_results=m.get("plot_reg_results", False),
                                        include_ee=m.get("include_ee_metric", False), **common)
    else:
        model = PrithviSegmentationModule(num_classes=m["num_classes"], class_weights=t["class_weights"], **common)
    if not train_mode:
        ckpt = cfg.get("checkpoint_path")
        if not ckpt or str(ckpt) == "None":
            raise RuntimeError("checkpoint_path is required for eval / chip_inference")
        sd = torch.load(ckpt, map_location="cpu")["state_dict"]
        model.load_checkpoint_state_dict(sd, strict=True)
    return model
