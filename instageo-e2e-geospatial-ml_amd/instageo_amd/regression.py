"""Regression module on the same network (reference: ``instageo/model/regression.py:34-343``).

``PrithviRegressionModule`` = the Prithvi encoder + decode head with ONE output channel, masked MSE loss
(``labels != ignore_index``), optional ``log1p`` label scale, and streaming RMSE / MAE / R2 / Pearson / expected-error
metrics.  Loss, its gradient and the nine running metric sums come from one kernel (``ig_mse_loss``); everything else
(forward, backward, AdamW, data parallelism, checkpoint layout) is the segmentation module's fused path.
"""
from __future__ import annotations

from typing import Any, List, Optional

import torch

from . import ops
from .metrics import RunningRegressionMetrics
from .segmentation import PrithviSegmentationModule

__all__ = ["LogScaler", "PrithviRegressionModule", "PrithviDistillationRegressionModule"]


class LogScaler:
    """log1p / expm1 label scale (regression.py:34-61)."""

    def transform(self, x: torch.Tensor) -> torch.Tensor:
        return torch.log1p(x)

    def inverse_transform(self, x: torch.Tensor) -> torch.Tensor:
        return torch.expm1(x)


class _RegLoss(torch.autograd.Function):
    """mean over the valid pixels of (pred - label')^2 [+ (pred - teacher')^2 with a teacher] through the fused kernels; the
    gradient w.r.t. the prediction is what ``ig_mse_loss`` [+ ``ig_kd_mse_loss``] wrote, divided by the valid count."""

    @staticmethod
    def forward(ctx, outputs, labels, teacher, ignore_index, use_log_scale, metrics):
        stats = torch.zeros(2, dtype=torch.float64, device=outputs.device)
        kd = torch.zeros(1, dtype=torch.float64, device=outputs.device)
        dl = torch.empty_like(outputs)
        lab = labels.to(torch.float32).contiguous()
        ops.mse_loss(outputs.contiguous(), lab, float(ignore_index), use_log_scale, stats, dl, metrics.device_sums(outputs.device),
                     metrics.ee_bias, metrics.ee_coef, metrics.include_ee)
        if teacher is not None:
            ops.kd_mse_loss(outputs.contiguous(), teacher.contiguous(), lab, float(ignore_index), use_log_scale, kd, dl)
        ctx.save_for_backward(dl, stats)
        n = stats[1]
        return ((stats[0] + kd[0]) / n).float(), (stats[0] / n).float().detach(), (kd[0] / n).float().detach()

    @staticmethod
    def backward(ctx, g, _g_mse, _g_kd):
        dl, stats = ctx.saved_tensors
        return dl * (g / stats[1].float()), None, None, None, None, None


class PrithviRegressionModule(PrithviSegmentationModule):
    """Same constructor surface as the reference (regression.py:67-131); ``num_classes`` is fixed to 1."""

    def __init__(self, image_size: int = 224, learning_rate: float = 1e-4, freeze_backbone: bool = True,
                 load_pretrained_weights: bool = True, temporal_step: int = 1, ignore_index: int = -100, weight_decay: float = 1e-2,
                 scheduler: bool = True, model_name: str = "prithvi_eo_v1_100", use_log_scale: bool = False,
                 plot_reg_results: bool = False, include_ee: bool = False, weight_clip_range: Optional[List[float]] = None,
                 depth: int = -1, precision: str = "bf16", device: Optional[Any] = None, **kwargs: Any) -> None:
        super().__init__(image_size=image_size, learning_rate=learning_rate, freeze_backbone=freeze_backbone,
                         load_pretrained_weights=load_pretrained_weights, num_classes=1, temporal_step=temporal_step,
                         class_weights=None, ignore_index=ignore_index, weight_decay=weight_decay, scheduler=scheduler,
                         model_name=model_name, weight_clip_range=weight_clip_range, depth=depth, precision=precision, device=device)
        if plot_reg_results:
            raise NotImplementedError("plot_reg_results (matplotlib/seaborn scatter plots) is outside the hot path")
        self.use_log_scale = use_log_scale
        self.include_ee = include_ee
        self.log_scaler = LogScaler()
        self.train_metrics = RunningRegressionMetrics(include_ee=include_ee)
        self.val_metrics = RunningRegressionMetrics(include_ee=include_ee)
        self.test_metrics = RunningRegressionMetrics(include_ee=include_ee)

    @property
    def num_classes(self) -> int:
        return 1

    def _fused_loss(self, logits, labels, stats, dlogits, step_type: str) -> None:
        """outputs.squeeze(1)[mask] vs labels[mask] (log1p-scaled when use_log_scale): MSE mean, metrics on the de-scaled
        values (regression.py:153-174) -- one kernel; ``stats`` = (sum of squared errors, #valid)."""
        metrics: RunningRegressionMetrics = getattr(self, f"{step_type}_metrics")
        ops.mse_loss(logits, labels.to(torch.float32).contiguous(), float(self.ignore_index), self.use_log_scale, stats, dlogits,
                     metrics.device_sums(logits.device), metrics.ee_bias, metrics.ee_coef, metrics.include_ee)

    def _teacher_outputs(self, inputs: torch.Tensor) -> Optional[torch.Tensor]:
        return None  # the distillation subclass runs its frozen teacher here

    def _shared_step(self, batch: Any, step_type: str) -> torch.Tensor:
        """Compatible (Lightning-style) path of regression.py:141-191: forward through the autograd bridge, masked MSE and the
        streaming metrics from the fused kernel; ``training_step(...).backward()`` then works like the reference's."""
        inputs, labels = batch
        outputs = self.forward(inputs)
        metrics = getattr(self, f"{step_type}_metrics")
        loss, mse, kd = _RegLoss.apply(outputs, labels, self._teacher_outputs(inputs), self.ignore_index, self.use_log_scale, metrics)
        if self._teacher_outputs.__func__ is not PrithviRegressionModule._teacher_outputs:
            self.log(f"{step_type}_mse_loss", mse.item())
            self.log(f"{step_type}_distill_loss", kd.item())
        self._accumulate_loss(step_type, loss.detach())
        return loss

    def _shared_epoch_end(self, step_type: str) -> None:
        metrics = getattr(self, f"{step_type}_metrics")
        m = metrics.compute()
        acc = self._loss_sums.pop(step_type, None)
        if acc is not None:
            self.log(f"{step_type}_loss", (acc[0] / acc[1]).item())
        self.log(f"{step_type}_RMSE", m["rmse"])
        self.log(f"{step_type}_MAE", m["mae"])
        self.log(f"{step_type}_R2", m["r2_score"])
        self.log(f"{step_type}_Pearson", m["pearson_corrcoef"])
        if m["ee_percentage"] is not None:
            self.log(f"{step_type}_EE_Percentage", m["ee_percentage"])
        metrics.reset()

    def predict_step(self, batch: Any) -> torch.Tensor:
        """forward(batch).squeeze(1), de-scaled when use_log_scale (regression.py:329-342)."""
        inputs = batch[0] if isinstance(batch, (tuple, list)) else batch
        if self.net.training:
            self.net.eval()
        with torch.no_grad():
            pred = self.net.engine.forward(inputs, training=False, save=False).squeeze(1)
        return self.log_scaler.inverse_transform(pred) if self.use_log_scale else pred


class PrithviDistillationRegressionModule(PrithviRegressionModule):
    """Knowledge distillation of the regression task (reference: ``regression.py:345-534``, ``base.py:234-334``): a frozen
    ``PrithviRegressionModule`` teacher of ``depth`` blocks loaded from ``teacher_ckpt_path`` and a student of ``student_depth``
    blocks; loss = mean((student - label')^2) + mean((student - teacher')^2) over the valid pixels, label' / teacher' =
    ``log1p`` of the label / teacher output under ``use_log_scale`` (regression.py:522-534, 496-503).  The second term and its
    gradient come from ``ig_kd_mse_loss`` on top of the fused ``ig_mse_loss``; the teacher runs the engine's inference path.
    Like the segmentation variant, ``student_depth`` is explicit (the reference passes it through ``**kwargs``)."""

    def __init__(self, teacher_ckpt_path: str, image_size: int = 224, learning_rate: float = 1e-4, load_pretrained_weights: bool = True,
                 temporal_step: int = 1, ignore_index: int = -100, weight_decay: float = 1e-2, scheduler: bool = True, depth: int = -1,
                 student_depth: int = -1, model_name: str = "prithvi_eo_v1_100", use_log_scale: bool = False,
                 plot_reg_results: bool = False, include_ee: bool = False, weight_clip_range: Optional[List[float]] = None,
                 freeze_backbone: bool = False, precision: str = "bf16", device: Optional[Any] = None, **kwargs: Any) -> None:
        super().__init__(image_size=image_size, learning_rate=learning_rate, freeze_backbone=freeze_backbone,
                         load_pretrained_weights=False, temporal_step=temporal_step, ignore_index=ignore_index,
                         weight_decay=weight_decay, scheduler=scheduler, model_name=model_name, use_log_scale=use_log_scale,
                         plot_reg_results=plot_reg_results, include_ee=include_ee, weight_clip_range=weight_clip_range,
                         depth=student_depth, precision=precision, device=device)
        self.teacher = PrithviRegressionModule(image_size=image_size, learning_rate=learning_rate, freeze_backbone=True,
                                               load_pretrained_weights=False, temporal_step=temporal_step, ignore_index=ignore_index,
                                               weight_decay=weight_decay, scheduler=scheduler, model_name=model_name, depth=depth,
                                               precision=precision, device=device)
        sd = torch.load(teacher_ckpt_path, map_location="cpu")["state_dict"]
        sd = {k.replace("prithvi_100M_backbone", "prithvi_encoder"): v for k, v in sd.items()}  # regression.py:461-464
        self.teacher.load_checkpoint_state_dict(sd, strict=True)
        self.teacher.net.eval()
        for p_ in self.teacher.parameters():
            p_.requires_grad_(False)
        if load_pretrained_weights:  # shared encoder tensors of equal shape initialise the student (base.py:312-324)
            t_sd = {k: v for k, v in self.teacher.net.state_dict().items() if k.startswith("prithvi_encoder.")}
            s_sd = self.net.state_dict()
            shared = {k: v for k, v in t_sd.items() if k in s_sd and v.shape == s_sd[k].shape}
            self.net.load_state_dict({**s_sd, **shared}, strict=True)
        self._kd = torch.zeros(1, dtype=torch.float64, device=self.net.store.flat.device)

    def _teacher_outputs(self, inputs: torch.Tensor) -> Optional[torch.Tensor]:
        with torch.no_grad():
            return self.teacher.net.engine.forward(inputs, training=False, save=False)

    def _fused_loss(self, logits, labels, stats, dlogits, step_type: str) -> None:
        super()._fused_loss(logits, labels, stats, dlogits, step_type)  # label term, its gradient, the regression metrics
        with torch.no_grad():
            t_out = self.teacher.net.engine.forward(self._last_inputs, training=False, save=False)
        self._kd.zero_()
        ops.kd_mse_loss(logits, t_out, labels.to(torch.float32).contiguous(), float(self.ignore_index), self.use_log_scale, self._kd, dlogits)
        self._parts = (stats[0].clone(), self._kd[0].clone(), stats[1].clone())
        stats[0] += self._kd[0]  # total = (sum sq. label error + sum sq. teacher error) / #valid

    def fused_train_step(self, inputs, labels, stats=None, grad_scale_world: int = 1):
        self._last_inputs = inputs
        out = super().fused_train_step(inputs, labels, stats, grad_scale_world)
        self._log_parts("train")
        return out

    def fused_eval_step(self, inputs, labels, step_type: str = "val"):
        self._last_inputs = inputs
        out = super().fused_eval_step(inputs, labels, step_type)
        self._log_parts(step_type)
        return out

    def _log_parts(self, step_type: str) -> None:
        mse, kd, n = self._parts
        self.log(f"{step_type}_mse_loss", (mse / n).item())
        self.log(f"{step_type}_distill_loss", (kd / n).item())
