"""Segmentation task module (reference: ``instageo/model/base.py:33-231`` + ``segmentation.py:33-213``).

``PrithviSegmentationModule`` keeps the reference's constructor arguments, step methods, metric names
and optimiser/scheduler choices.  Two execution paths share the same kernels:

* the *compatible* path -- ``loss = module.training_step(batch, i); loss.backward(); optimizer.step()`` --
  works under an unmodified (Lightning-style) loop through two ``torch.autograd.Function`` bridges;
* the *fused* path -- :meth:`fused_train_step` -- runs forward, loss+metrics, backward, (gradient
  all-reduce) and AdamW as one stream of HIP launches with no host synchronisation; this is what
  ``run.py``, ``bench.py`` and the data-parallel trainer use.
"""
from __future__ import annotations

import math
import os
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .metrics import RunningAUC, RunningConfusionMatrix
from .model import PrithviSeg

try:  # the module must still work as a LightningModule when Lightning is installed
    import pytorch_lightning as pl  # type: ignore

    _Base = pl.LightningModule
except Exception:  # pragma: no cover - Lightning is absent in this image
    pl = None
    _Base = nn.Module


# --------------------------------------------------------------------------------------------------
# AdamW on the flat parameter buffer (base.py:124-126: torch.optim.AdamW defaults)
# --------------------------------------------------------------------------------------------------
_FRESH_STEP = True     # the Blocks' weight gradients are written, not accumulated: the step neither zeroes nor reads them (round 3)
_ADAMW_OVERLAP = True  # AdamW per gradient range on a side stream during backward (round 4: +0.1-0.7 % at B = 216; off below ~50 chips)

class FusedAdamW(torch.optim.Optimizer):
    """``torch.optim.AdamW`` semantics, one HIP launch over the flat buffer of a :class:`PrithviSeg`.

    Subclasses ``torch.optim.Optimizer`` so LR schedulers (CosineAnnealingWarmRestarts, base.py:128-131)
    drive ``param_groups[0]["lr"]`` as usual.  Gradients are taken from the flat grad buffer when the
    fused path produced them, otherwise gathered from ``p.grad``.
    """

    def __init__(self, net: PrithviSeg, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 clip_range: Optional[List[float]] = None):
        params = [p for p in net.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.net = net
        self.clip_range = clip_range
        store = net.store
        self.lo = store.encoder_end if net.freeze_backbone else 0
        self.hi = store.total
        dev = store.flat.device
        self.m = torch.zeros(self.hi - self.lo, dtype=torch.float32, device=dev)
        self.v = torch.zeros_like(self.m)
        self.hyper = torch.zeros(16, dtype=torch.float32, device=dev)
        self.sharded = None  # distributed.ShardedGradSync when the optimizer state is sharded over the data-parallel ranks
        self._small_ranges = None  # flat ranges of the parameters the kernels read in fp32 (biases, norm affine, classifier ...)
        self._side = None  # side stream of the early (overlapped) optimizer launches
        self._host_step = 0
        self._write_hyper()

    def _write_hyper(self) -> None:
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        h = [g["lr"], b1, b2, g["eps"], g["weight_decay"], 0.0, 0.0, 0.0, 0.0, 0.0, float(self._host_step), 1 - b1, 1 - b2]
        if self.clip_range is not None:
            h[7], h[8], h[9] = float(self.clip_range[0]), float(self.clip_range[1]), 1.0
        self.hyper[: len(h)] = torch.tensor(h, dtype=torch.float32)
        self._lr_written = g["lr"]

    def attach_sharded(self, sync) -> None:
        """Shard the optimizer over the data-parallel ranks (``distributed.ShardedGradSync``): the replicated moment buffers
        are released, every rank keeps the moments of its slices only."""
        self.sharded = sync
        self.m = self.v = None

    def set_grad_scale(self, scale: float) -> None:
        """Multiply gradients by ``scale`` inside the AdamW kernel (1/world_size after a SUM all-reduce)."""
        self.hyper[13] = float(scale)

    def gather_grads(self) -> None:
        """Copy autograd-produced ``p.grad`` tensors into the flat grad buffer (compatible path)."""
        store = self.net.store
        g = store.ensure_grad()
        for name, p in self.net._flat_params():
            if p.requires_grad and p.grad is not None:
                store.entries[name].api_view(g).copy_(p.grad)

    def _begin(self):
        """Per-step preamble: learning rate, device step counter / bias corrections, operand copy in the engine's precision."""
        if self.param_groups[0]["lr"] != self._lr_written:
            self.hyper[0] = float(self.param_groups[0]["lr"])
            self._lr_written = self.param_groups[0]["lr"]
        store = self.net.store
        self._host_step += 1
        ops.adamw_advance(self.hyper)
        # the kernel also refreshes the bf16 (hi/lo) operand copy used by the MFMA kernels
        eng = self.net.engine
        if store.shadow is None or store.shadow_split != eng.split:
            if eng.master_sync is not None:  # sharded fp32 masters: complete them before the operand copy is rebuilt from them (collective)
                eng.master_sync()
            store.refresh_shadow(eng.split)
        return store, eng, store.shadow

    def _adam_range(self, lo: int, hi: int) -> None:
        """AdamW on flat range [lo, hi) (inside [self.lo, self.hi)) with the replicated moments."""
        store, sh = self.net.store, self.net.store.shadow
        shadow = ops.BT(sh.hi[lo:hi], None if sh.lo is None else sh.lo[lo:hi])
        ops.adamw_step(store.flat[lo:hi], store.grad[lo:hi], self.m[lo - self.lo : hi - self.lo], self.v[lo - self.lo : hi - self.lo], shadow,
                       self.hyper, hi - lo)

    @torch.no_grad()
    def step(self, closure=None, grads_in_flat: bool = False):
        loss = closure() if closure is not None else None
        if not grads_in_flat:
            self.gather_grads()
        store, eng, sh = self._begin()
        if self.sharded is not None:
            # data parallel, "zero1": the gradient buckets were reduce-scattered during backward; AdamW runs on this rank's slice
            # of every bucket (moments live in the ShardedGradSync) and writes the fp32 masters AND the bf16 operand copy of the
            # slice.  The other ranks only compute with the bf16 copy, so THAT is all-gathered (2 bytes per parameter on the wire
            # instead of 4, and no refresh pass over the slices other ranks updated); the fp32 masters stay sharded until
            # gather_master() (checkpoints), the fp32-read vectors travel in one small all-reduce.  (Round 3's form -- all-gather of the
            # fp32 parameters + a refresh of the whole operand copy -- was an A/B arm and is gone.)
            def adam(param, grad, m, v, index0):
                shw = ops.BT(sh.hi[index0 : index0 + param.numel()], None if sh.lo is None else sh.lo[index0 : index0 + param.numel()])
                ops.adamw_step(param, grad, m, v, shw, self.hyper, param.numel())

            if self._small_ranges is None:
                self._small_ranges = [(e.offset, e.offset + e.numel) for e in store.entries.values()
                                      if e.numel <= 65536 and e.offset + e.numel > self.lo and e.offset < self.hi]
            # the waits for the all-gathers are left to the next forward pass (engine.param_wait, one per Block): IG_DP_DEFER=0 waits here
            self.sharded.step(adam, gather=[sh.hi] + ([] if sh.lo is None else [sh.lo]), small_ranges=self._small_ranges,
                              defer=os.environ.get("IG_DP_DEFER", "1") != "0" and eng.param_wait is not None)
            eng.shadow_dirty = False
            eng.shadow_t_dirty = True
            return loss
        self._adam_range(self.lo, self.hi)
        eng.shadow_dirty = False
        eng.shadow_t_dirty = True  # the transposed weight copy is rebuilt by the next backward
        return loss

    # ---- single process: AdamW of every gradient range as soon as it is final, on a side stream ------------------------------------
    def early_begin(self) -> Callable[[int, int], None]:
        """Start a step whose optimizer work rides on the backward pass: returns the ``on_grad_ready(lo, hi)`` hook.  Ranges arrive
        adjacent, in descending address order (head first); every >= ``EARLY_MIN`` elements the AdamW kernel of the merged range is
        launched on a SIDE stream behind an event of the launch stream, so it overlaps the backward kernels of the earlier blocks
        (whose weights it does not touch: a block's parameters are last read by its own backward).  :meth:`early_finish` joins."""
        for hook in list(getattr(self, "_optimizer_step_pre_hooks", {}).values()):  # this path IS optimizer.step(): same hooks
            hook(self, (), {})
        self._begin()
        if self._side is None:
            self._side = torch.cuda.Stream()
        self._early_cur: Optional[Tuple[int, int]] = None
        self._early_done: List[Tuple[int, int]] = []
        return self._early_ready

    EARLY_MIN = 12 << 20  # elements per launch: a handful of launches per step (each notification below it only extends the pending range)

    def _early_ready(self, lo: int, hi: int) -> None:
        lo, hi = max(lo, self.lo), min(hi, self.hi)
        if hi <= lo:
            return
        if self._early_cur is not None and hi == self._early_cur[0]:
            self._early_cur = (lo, self._early_cur[1])
        else:
            self._early_flush()
            self._early_cur = (lo, hi)
        if self._early_cur[1] - self._early_cur[0] >= self.EARLY_MIN:
            self._early_flush()

    def _early_flush(self) -> None:
        if self._early_cur is None:
            return
        lo, hi = self._early_cur
        self._early_cur = None
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            self._adam_range(lo, hi)
        self._early_done.append((lo, hi))

    @torch.no_grad()
    def early_finish(self) -> None:
        """Launch what is left, make the launch stream wait for the side stream, and update whatever was never reported."""
        self._early_flush()
        torch.cuda.current_stream().wait_stream(self._side)
        at = self.hi
        for lo, hi in self._early_done:  # descending, adjacent when the whole range was reported
            if hi < at:
                self._adam_range(hi, at)
            at = min(at, lo)
        if at > self.lo:
            self._adam_range(self.lo, at)
        eng = self.net.engine
        eng.shadow_dirty = False
        eng.shadow_t_dirty = True
        # what torch.optim.Optimizer.step()'s wrappers would have done: LR schedulers check _opt_called ("lr_scheduler.step() before
        # optimizer.step()"), step post hooks run after the update
        self._opt_called = True
        for hook in list(getattr(self, "_optimizer_step_post_hooks", {}).values()):
            hook(self, (), {})

    def early_abort(self) -> None:
        """The backward pass raised in the middle of an overlapped step: join the side stream (its launches read the gradient
        buffer) and take back the step counter the preamble advanced.  The BOOKKEEPING is rolled back; the PARAMETERS are not: the AdamW
        launches that were already issued (head and late blocks) have updated their fp32 masters and moments, so a failed overlapped step
        leaves the model partially updated and must not be retried as if it had not happened (a retry would apply those ranges twice with
        the same bias-correction step) -- restore a checkpoint instead, as after any exception inside ``optimizer.step()``.  Step pre / post
        hooks registered on this optimizer are called with empty ``(args, kwargs)`` and their return values are ignored; the global hooks of
        ``torch.optim.Optimizer.step``'s wrapper do not run on the overlapped path."""
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._early_cur, self._early_done = None, []
        self._host_step -= 1
        self._write_hyper()
        self.net.engine.mark_params_changed()  # some ranges may already be updated: rebuild the operand copy from the masters

    def zero_grad(self, set_to_none: bool = True) -> None:
        super().zero_grad(set_to_none=set_to_none)
        if self.net.store.grad is not None:
            self.net.store.grad.zero_()


# --------------------------------------------------------------------------------------------------
# loss bridge: CrossEntropyLoss(weight, ignore_index, 'none') + loss[mask].mean()  (segmentation.py:85-87,117-122)
# --------------------------------------------------------------------------------------------------
class _SegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, weight, ignore_index, confusion, preds):
        stats = torch.zeros(2, dtype=torch.float64, device=logits.device)
        dlog = torch.empty_like(logits)
        ops.ce_loss(logits.contiguous(), labels.contiguous(), weight, ignore_index, stats, dlog, preds, None, confusion)
        ctx.save_for_backward(dlog, stats)
        return (stats[0] / stats[1]).float()

    @staticmethod
    def backward(ctx, g):
        dlog, stats = ctx.saved_tensors
        return dlog * (g / stats[1].float()), None, None, None, None, None


def segmentation_loss(logits: torch.Tensor, labels: torch.Tensor, class_weights: Optional[torch.Tensor], ignore_index: int,
                      confusion: Optional[torch.Tensor] = None, preds: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sum_valid(w_y * nll) / #valid -- NOT torch's weighted mean.  NaN when nothing is valid,
    exactly like ``loss[mask].mean()`` of an empty selection."""
    if labels.dtype not in (torch.int64, torch.int32, torch.float32):
        labels = labels.long()
    return _SegLoss.apply(logits, labels, class_weights, ignore_index, confusion, preds)


class _SegKDLoss(torch.autograd.Function):
    """CE(student) + KLDivLoss(batchmean)(log_softmax(student), softmax(teacher)) over the valid pixels
    (segmentation.py:352-378): total, and the two parts for logging."""

    @staticmethod
    def forward(ctx, logits, t_logits, labels, weight, ignore_index, confusion):
        stats = torch.zeros(2, dtype=torch.float64, device=logits.device)
        kl = torch.zeros(1, dtype=torch.float64, device=logits.device)
        dlog = torch.empty_like(logits)
        ops.ce_loss(logits.contiguous(), labels.contiguous(), weight, ignore_index, stats, dlog, None, None, confusion)
        ops.kd_loss(logits.contiguous(), t_logits.contiguous(), labels.contiguous(), ignore_index, kl, dlog)
        ctx.save_for_backward(dlog, stats)
        return ((stats[0] + kl[0]) / stats[1]).float(), (stats[0] / stats[1]).float().detach(), (kl[0] / stats[1]).float().detach()

    @staticmethod
    def backward(ctx, g, _g_ce, _g_kl):
        dlog, stats = ctx.saved_tensors
        return dlog * (g / stats[1].float()), None, None, None, None, None


# --------------------------------------------------------------------------------------------------
class PrithviSegmentationModule(_Base):
    """Prithvi Segmentation module with the reference's constructor (segmentation.py:36-98)."""

    def __init__(
        self,
        image_size: int = 224,
        learning_rate: float = 1e-4,
        freeze_backbone: bool = True,
        load_pretrained_weights: bool = True,
        num_classes: int = 2,
        temporal_step: int = 1,
        class_weights: Optional[List[float]] = None,
        ignore_index: int = -100,
        weight_decay: float = 1e-2,
        scheduler: bool = True,
        model_name: str = "prithvi_eo_v1_100",
        weight_clip_range: Optional[List[float]] = None,
        depth: int = -1,
        precision: str = "bf16",
        device: Optional[Any] = None,
    ) -> None:
        super().__init__()
        self._num_classes = num_classes
        self.net = PrithviSeg(
            image_size=image_size, temporal_step=temporal_step, freeze_backbone=freeze_backbone, variant=model_name,
            load_pretrained_weights=load_pretrained_weights, depth=depth, num_classes=num_classes, precision=precision, device=device,
        )  # fmt: skip
        self.learning_rate = learning_rate
        self.weight_decay = weight_decay
        self.scheduler = scheduler
        self.weight_clip_range = weight_clip_range
        self.ignore_index = ignore_index
        dev = self.net.store.flat.device
        # `criterion.weight` is part of the reference checkpoint: keep the same buffer path
        self.criterion = nn.Module()
        if class_weights:
            self.criterion.register_buffer("weight", torch.tensor(class_weights).float().to(dev))
        else:
            self.criterion.weight = None
        self.train_metrics = RunningConfusionMatrix(num_classes, ignore_index)
        self.val_metrics = RunningConfusionMatrix(num_classes, ignore_index)
        self.test_metrics = RunningConfusionMatrix(num_classes, ignore_index)
        self.test_auc = RunningAUC(num_classes, ignore_index=ignore_index)  # ROC-AUC only at test time (segmentation.py:153-156)
        self.logged: Dict[str, Any] = {}
        self._loss_sums: Dict[str, torch.Tensor] = {}
        self._optimizer: Optional[FusedAdamW] = None
        self._early_ok = True  # False while a hipGraph of the step is being prepared (make_graphed_train_step)
        self.grad_sync: Optional[Callable[[], None]] = None  # set by the data-parallel wrapper

    # ---- reference API -------------------------------------------------------------------------
    @property
    def num_classes(self) -> int:
        return self._num_classes

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.net(x)

    def log(self, name: str, value: Any, **kw: Any) -> None:  # Lightning's self.log when available
        if pl is not None and getattr(self, "_trainer", None) is not None:  # pragma: no cover
            return super().log(name, value, **kw)
        self.logged[name] = value

    def clip_weights(self) -> None:
        """base.py:103-113 -- in the fused path the clamp runs inside the AdamW kernel instead."""
        if self.weight_clip_range is not None:
            lo, hi = self.weight_clip_range
            with torch.no_grad():
                self.net.store.flat.clamp_(lo, hi)
                self.net.params_changed()

    def configure_optimizers(self):
        opt = FusedAdamW(self.net, lr=self.learning_rate, weight_decay=self.weight_decay, clip_range=None)
        self._optimizer = opt
        if self.scheduler:
            sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2, eta_min=0)
            return [opt], [sch]
        return [opt], []

    def _weights(self) -> Optional[torch.Tensor]:
        return getattr(self.criterion, "weight", None)

    def _shared_step(self, batch: Any, step_type: str) -> torch.Tensor:
        """forward, loss, argmax, confusion-matrix update (segmentation.py:107-168) without host copies."""
        inputs, labels = batch
        outputs = self.forward(inputs)
        metrics: RunningConfusionMatrix = getattr(self, f"{step_type}_metrics")
        loss = segmentation_loss(outputs, labels, self._weights(), self.ignore_index, confusion=metrics.device_matrix(outputs.device))
        if step_type == "test":  # ROC-AUC is a test-time metric (segmentation.py:153-156)
            self.test_auc.update_from_logits(outputs.detach(), labels)
        self._accumulate_loss(step_type, loss.detach())
        return loss

    def _accumulate_loss(self, step_type: str, loss: torch.Tensor) -> None:
        acc = self._loss_sums.get(step_type)
        if acc is None:
            acc = torch.zeros(2, dtype=torch.float64, device=loss.device)
            self._loss_sums[step_type] = acc
        acc[0] += loss.double()
        acc[1] += 1

    def training_step(self, batch: Any, batch_idx: int = 0) -> torch.Tensor:
        loss = self._shared_step(batch, "train")
        if self._optimizer is not None:
            self.log("learning_rate", self._optimizer.param_groups[0]["lr"])
        self.clip_weights()
        return loss

    def validation_step(self, batch: Any, batch_idx: int = 0) -> torch.Tensor:
        with torch.no_grad():
            return self._shared_step(batch, "val")

    def test_step(self, batch: Any, batch_idx: int = 0) -> torch.Tensor:
        with torch.no_grad():
            return self._shared_step(batch, "test")


    def _shared_epoch_end(self, step_type: str) -> None:
        metrics = getattr(self, f"{step_type}_metrics")
        m = metrics.compute()
        acc = self._loss_sums.pop(step_type, None)
        if acc is not None:
            self.log(f"{step_type}_loss", (acc[0] / acc[1]).item())
        self.log(f"{step_type}_Acc", m["accuracy"])
        self.log(f"{step_type}_IoU", m["jaccard"])
        self.log(f"{step_type}_F1", m["f1"])
        self.log(f"{step_type}_Precision", m["precision"])
        self.log(f"{step_type}_Recall", m["recall"])
        for idx, value in enumerate(m["jaccard_per_class"]):
            self.log(f"{step_type}_IoU_{idx}", value)
        if step_type == "test":
            self.log(f"{step_type}_roc_auc", float(self.test_auc.score()["roc_auc_macro"]))
        for idx, value in enumerate(m["f1_per_class"]):
            self.log(f"{step_type}_F1_{idx}", value)
        metrics.reset()
        if step_type == "test":
            self.test_auc.reset()

    def on_train_epoch_end(self) -> None:
        self._shared_epoch_end("train")

    def on_validation_epoch_end(self) -> None:
        self._shared_epoch_end("val")

    def on_test_epoch_end(self) -> None:
        self._shared_epoch_end("test")

    # ---- fused path ----------------------------------------------------------------------------
    def optimizer(self) -> FusedAdamW:
        if self._optimizer is None:
            self._optimizer = FusedAdamW(self.net, lr=self.learning_rate, weight_decay=self.weight_decay, clip_range=self.weight_clip_range)
        return self._optimizer

    def fused_train_step(self, inputs: torch.Tensor, labels: torch.Tensor, stats: Optional[torch.Tensor] = None,
                         grad_scale_world: int = 1) -> torch.Tensor:
        """One full training step (forward, loss+metrics, backward, all-reduce hook, AdamW) with no host sync.

        Returns the device double[2] = (sum of weighted nll, #valid pixels) of this batch; loss = [0]/[1].
        """
        net, eng = self.net, self.net.engine
        opt = self.optimizer()
        if not net.training:
            net.train()
        if labels.dtype not in (torch.int64, torch.int32, torch.float32):
            labels = labels.long()
        logits = eng.forward(inputs, training=True, save=True)
        if stats is None:
            stats = torch.zeros(2, dtype=torch.float64, device=logits.device)
        else:
            stats.zero_()
        ws = eng._last["ws"]
        dlog = ws.get("dlogits")
        if dlog is None or dlog.shape != logits.shape:
            dlog = torch.empty_like(logits)
            ws["dlogits"] = dlog
        self._fused_loss(logits, labels, stats, dlog, "train")
        g = net.store.ensure_grad()
        if _FRESH_STEP:  # the Blocks' weight gradients are written, not accumulated: they are neither zeroed nor read (model.py)
            eng.zero_grads_for_step(opt.lo, opt.hi)
        else:
            g[opt.lo : opt.hi].zero_()
        # single process, not under a stream capture: the optimizer work rides on the backward pass
        # (measured, tools/ab_step.sh: +0.5-0.7 % at B = 216; at the YAML's batch 16 the extra fold launches and stream events cost
        # more than the 0.5 ms they hide -- the step there is bound by the host's launch rate -- so small batches keep the single launch)
        early = (self.grad_sync is None and opt.sharded is None and eng.on_grad_ready is None and _ADAMW_OVERLAP and self._early_ok
                 and inputs.shape[0] * net.cfg.tokens >= 10000 and not torch.cuda.is_current_stream_capturing())
        if early:
            eng.on_grad_ready = opt.early_begin()
            try:
                eng.backward(dlog, count=stats, fresh=_FRESH_STEP)
            except BaseException:
                eng.on_grad_ready = None
                opt.early_abort()
                raise
            eng.on_grad_ready = None
            opt.early_finish()
        else:
            eng.backward(dlog, count=stats, fresh=_FRESH_STEP)
            if self.grad_sync is not None:
                self.grad_sync()
            opt.step(grads_in_flat=True)
        acc = self._loss_sums.get("train")
        if acc is None:
            acc = torch.zeros(2, dtype=torch.float64, device=logits.device)
            self._loss_sums["train"] = acc
        acc[0] += stats[0] / stats[1]
        acc[1] += 1
        return stats

    def make_graphed_train_step(self, inputs: torch.Tensor, labels: torch.Tensor):
        """Capture :meth:`fused_train_step` into one hipGraph (single-process only: no collectives inside).

        Returns ``run(inputs, labels) -> stats``: copies the batch into the graph's static buffers and replays ~250
        kernel launches with one host call.  Everything the step needs is device resident (AdamW step counter and
        bias corrections, dropout seed counter, loss statistics), so replays need no host-side scalars.
        """
        assert self.grad_sync is None and self.optimizer().sharded is None, "graph capture is for the single-GPU path (RCCL calls are not captured)"
        static_x = inputs.clone()
        static_y = labels.clone()
        stats = torch.zeros(2, dtype=torch.float64, device=inputs.device)
        # The warm-up below runs two REAL optimizer steps and the capture pass itself must not count as training either:
        # snapshot everything a step mutates (parameters, AdamW moments + step counter, BatchNorm running statistics,
        # dropout counter, streaming metrics, loss accumulator) and restore it afterwards, so that capturing is free of
        # side effects and the first replay is training step 1.
        opt = self.optimizer()
        eng = self.net.engine
        eng._drop_counter(advance=False)
        self.train_metrics.device_matrix(inputs.device)
        if "train" not in self._loss_sums:
            self._loss_sums["train"] = torch.zeros(2, dtype=torch.float64, device=inputs.device)
        snap = {"flat": self.net.store.flat.clone(), "m": opt.m.clone(), "v": opt.v.clone(), "hyper": opt.hyper.clone(),
                "host_step": opt._host_step, "drop": eng._drop_step.clone(), "cm": self.train_metrics.device_matrix().clone(),
                "loss": self._loss_sums["train"].clone(), "bufs": {k: t.clone() for k, t in self.net._buffers_flat.items()}}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self._early_ok = False  # warm-up and capture run the SAME launch sequence (the overlapped optimizer forks a second stream)
        try:
            with torch.cuda.stream(side):  # warm-up on a side stream: allocates workspaces, sets kernel attributes
                for _ in range(2):
                    self.fused_train_step(static_x, static_y, stats)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # Captured on the warm-up's OWN stream: the library's scratch buffers are per (device, stream) and nothing can be allocated during
            # a capture, so the capture must meet the buffers the warm-up made.  The graph's kernels keep using that stream's buffers wherever
            # it is replayed, so a replay never shares scratch with eager work on the replaying stream either.
            with torch.cuda.graph(graph, stream=side):
                self.fused_train_step(static_x, static_y, stats)
        finally:
            self._early_ok = True
        with torch.no_grad():  # in-place restores: the graph keeps writing into these very tensors
            self.net.store.flat.copy_(snap["flat"])
            opt.m.copy_(snap["m"]), opt.v.copy_(snap["v"]), opt.hyper.copy_(snap["hyper"])
            opt._host_step = snap["host_step"]
            eng._drop_step.copy_(snap["drop"])
            self.train_metrics.device_matrix().copy_(snap["cm"])
            self._loss_sums["train"].copy_(snap["loss"])
            for k, t in self.net._buffers_flat.items():
                t.copy_(snap["bufs"][k])
            self.net.params_changed()
            eng._prepare_shadow()
        loss_acc = self._loss_sums["train"]  # the captured kernels accumulate into THIS tensor: keep it alive across epochs

        def run(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
            if opt.param_groups[0]["lr"] != opt._lr_written:  # scheduler changed the LR: one scalar write, outside the graph
                opt.hyper[0] = float(opt.param_groups[0]["lr"])
                opt._lr_written = opt.param_groups[0]["lr"]
            static_x.copy_(x)
            static_y.copy_(y)
            if self._loss_sums.get("train") is not loss_acc:  # epoch end popped the accumulator: re-install it, zeroed
                loss_acc.zero_()
                self._loss_sums["train"] = loss_acc
            graph.replay()
            opt._host_step += 1
            return stats

        run.graph = graph
        return run

    @torch.no_grad()
    def fused_eval_step(self, inputs: torch.Tensor, labels: torch.Tensor, step_type: str = "val") -> torch.Tensor:
        net, eng = self.net, self.net.engine
        if net.training:
            net.eval()
        logits = eng.forward(inputs, training=False, save=False)
        stats = torch.zeros(2, dtype=torch.float64, device=logits.device)
        if labels.dtype not in (torch.int64, torch.int32, torch.float32):
            labels = labels.long()
        self._fused_loss(logits, labels, stats, None, step_type)
        self._accumulate_loss(step_type, (stats[0] / stats[1]).float())
        return stats

    def _fused_loss(self, logits: torch.Tensor, labels: torch.Tensor, stats: torch.Tensor, dlogits: Optional[torch.Tensor],
                    step_type: str) -> None:
        """Loss statistics (+ un-normalised dlogits when training) and the step's streaming metrics, all on the device.
        Overridden by the regression module."""
        metrics: RunningConfusionMatrix = getattr(self, f"{step_type}_metrics")
        ops.ce_loss(logits, labels.contiguous(), self._weights(), self.ignore_index, stats, dlogits, None, None,
                    metrics.device_matrix(logits.device))
        if step_type == "test":
            self.test_auc.update_from_logits(logits, labels)

    def predict_step(self, batch: Any) -> torch.Tensor:
        """softmax(forward(batch), dim=1)[:, 1] (segmentation.py:202-213), fused on the device."""
        inputs = batch[0] if isinstance(batch, (tuple, list)) else batch
        if self.net.training:
            self.net.eval()
        with torch.no_grad():
            logits = self.net.engine.forward(inputs, training=False, save=False)
        return ops.softmax_prob(logits, 1)

    # ---- checkpoints (pipeline_utils.py:347-355, factory.py:113-115) ----------------------------
    def sync_master_params(self) -> None:
        """Data parallel (``zero1``): complete the fp32 master parameters on every rank -- between checkpoints only the bf16 operand
        copy is exchanged (``distributed.ShardedGradSync.gather_master``).  Collective: every rank calls it; a no-op otherwise."""
        opt = getattr(self, "_optimizer", None)
        sync = getattr(opt, "sharded", None) if opt is not None else None
        if sync is not None:
            sync.gather_master()

    def checkpoint_state_dict(self) -> Dict[str, torch.Tensor]:
        """``{"net.<...>": tensor, "criterion.weight": tensor}`` -- the reference's Lightning key layout."""
        sd = {"net." + k: v.detach().clone().contiguous().cpu() for k, v in self.net.state_dict().items()}
        if self._weights() is not None:
            sd["criterion.weight"] = self._weights().detach().cpu()
        return sd

    def load_checkpoint_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        net_sd = {k[len("net.") :]: v for k, v in sd.items() if k.startswith("net.")}
        self.net.load_state_dict(net_sd, strict=strict)
        if "criterion.weight" in sd and self._weights() is not None:
            self.criterion.weight.copy_(sd["criterion.weight"])


class PrithviDistillationSegmentationModule(PrithviSegmentationModule):
    """Knowledge distillation (reference: ``segmentation.py:216-451``, ``base.py:234-334``): a frozen teacher of ``depth``
    blocks loaded from ``teacher_ckpt_path`` and a student (normally shallower, ``student_depth``); loss = CE(student)
    + KLDivLoss(batchmean)(log_softmax(student), softmax(teacher)) over the valid pixels.  The KL term and its gradient come
    from ``ig_kd_loss`` on top of the fused CE kernel; the teacher runs the inference path of the same engine.

    Differences from the reference surface: ``student_depth`` is explicit (the reference passes the student depth through
    ``**kwargs`` of ``PrithviSeg``); the student's head keeps this package's seeded initialisation.
    """

    def __init__(self, teacher_ckpt_path: str, image_size: int = 224, learning_rate: float = 1e-4, num_classes: int = 2,
                 temporal_step: int = 1, class_weights: Optional[List[float]] = None, ignore_index: int = -100,
                 weight_decay: float = 1e-2, model_name: str = "prithvi_eo_v1_100", depth: int = -1, student_depth: int = -1,
                 load_pretrained_weights: bool = True, scheduler: bool = True, weight_clip_range: Optional[List[float]] = None,
                 freeze_backbone: bool = False, precision: str = "bf16", device: Optional[Any] = None, **kwargs: Any) -> None:
        super().__init__(image_size=image_size, learning_rate=learning_rate, freeze_backbone=freeze_backbone,
                         load_pretrained_weights=False, num_classes=num_classes, temporal_step=temporal_step,
                         class_weights=class_weights, ignore_index=ignore_index, weight_decay=weight_decay, scheduler=scheduler,
                         model_name=model_name, weight_clip_range=weight_clip_range, depth=student_depth, precision=precision,
                         device=device)
        self.teacher = PrithviSegmentationModule(image_size=image_size, learning_rate=learning_rate, freeze_backbone=True,
                                                 load_pretrained_weights=False, num_classes=num_classes, temporal_step=temporal_step,
                                                 class_weights=class_weights, ignore_index=ignore_index, weight_decay=weight_decay,
                                                 scheduler=scheduler, model_name=model_name, depth=depth, precision=precision,
                                                 device=device)
        sd = torch.load(teacher_ckpt_path, map_location="cpu")["state_dict"]
        # old checkpoints name the encoder "prithvi_100M_backbone" (segmentation.py:333-337)
        sd = {k.replace("prithvi_100M_backbone", "prithvi_encoder"): v for k, v in sd.items()}
        self.teacher.load_checkpoint_state_dict(sd, strict=True)
        self.teacher.net.eval()
        for p_ in self.teacher.parameters():
            p_.requires_grad_(False)
        if load_pretrained_weights:  # shared encoder tensors of equal shape are copied into the student (base.py:312-324)
            t_sd = {k: v for k, v in self.teacher.net.state_dict().items() if k.startswith("prithvi_encoder.")}
            s_sd = self.net.state_dict()
            shared = {k: v for k, v in t_sd.items() if k in s_sd and v.shape == s_sd[k].shape}
            self.net.load_state_dict({**s_sd, **shared}, strict=True)
        self._kl = torch.zeros(1, dtype=torch.float64, device=self.net.store.flat.device)

    def _shared_step(self, batch: Any, step_type: str) -> torch.Tensor:
        """Compatible (Lightning-style) path of the reference's ``_shared_step`` (segmentation.py:380-451): student forward
        through autograd, teacher forward under no_grad, loss = CE + KLDiv(batchmean); logs ``<step>_ce_loss`` and
        ``<step>_distill_loss`` like the fused path."""
        inputs, labels = batch
        outputs = self.forward(inputs)
        with torch.no_grad():
            t_logits = self.teacher.net.engine.forward(inputs, training=False, save=False)
        if labels.dtype not in (torch.int64, torch.int32, torch.float32):
            labels = labels.long()
        metrics: RunningConfusionMatrix = getattr(self, f"{step_type}_metrics")
        loss, ce, kl = _SegKDLoss.apply(outputs, t_logits, labels, self._weights(), self.ignore_index, metrics.device_matrix(outputs.device))
        if step_type == "test":
            self.test_auc.update_from_logits(outputs.detach(), labels)
        self.log(f"{step_type}_ce_loss", ce.item())
        self.log(f"{step_type}_distill_loss", kl.item())
        self._accumulate_loss(step_type, loss.detach())
        return loss

    def _fused_loss(self, logits, labels, stats, dlogits, step_type: str) -> None:
        super()._fused_loss(logits, labels, stats, dlogits, step_type)  # CE statistics, CE gradient, confusion matrix, AUC
        with torch.no_grad():
            t_logits = self.teacher.net.engine.forward(self._last_inputs, training=False, save=False)
        self._kl.zero_()
        ops.kd_loss(logits, t_logits, labels.contiguous(), self.ignore_index, self._kl, dlogits)
        # total = (ce_sum + kl_sum) / #valid: fold the KL numerator into the loss statistic; keep the parts for logging
        self._parts = (stats[0].clone(), self._kl[0].clone(), stats[1].clone())
        stats[0] += self._kl[0]

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
        ce, kl, n = self._parts
        self.log(f"{step_type}_ce_loss", (ce / n).item())
        self.log(f"{step_type}_distill_loss", (kl / n).item())
