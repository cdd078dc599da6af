"""CPU oracle for the Prithvi segmentation hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch / numpy *restatement* of the reference algorithm for the
path named by BASELINE.json:north_star.  It is the checker for the HIP product path:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product package (``instageo_amd``) never imports anything under
``oracle/`` and fails loudly when its HIP library is missing.

Pinning status
--------------
* network forward/backward: pinned against an import of the reference
  (``/root/reference/instageo/model/{model,pritvhi}.py``) in the build container by
  ``oracle/gen_golden.py``; the resulting vectors live in ``tests/golden/*.npz``.
  The transformer block itself is third-party (timm 1.0.20, ``uv.lock:5358``) and is
  *not* vendored in the reference: its arithmetic is restated from timm's documented
  ``Block``/``Attention``/``Mlp`` forward (pre-LN, qkv_bias, no LayerScale, exact-erf
  GELU, LN eps 1e-5 at the call site ``pritvhi.py:446-456``) -- that part is
  **parity unpinned** by any reference test.
* metrics: pinned against ``instageo/model/metrics.py`` (importable as-is) and its
  known answers (``tests/model_tests/test_metrics.py``).

Every function cites the reference file:line it follows (paths relative to the
reference root).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Configuration (instageo/model/model.py:39-177)
# --------------------------------------------------------------------------------------


@dataclass
class OracleConfig:
    """Architecture numbers of one reference variant (model.py:128-177)."""

    variant: str = "prithvi_eo_v1_100"
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: int = 4
    patch: int = 16
    in_chans: int = 6
    num_frames: int = 1  # temporal_step
    img_size: int = 224
    num_classes: int = 2
    head_kernels: Tuple[int, int, int, int] = (3, 3, 3, 3)
    embed_dims: Optional[Tuple[int, ...]] = None  # PrithviSeg(embed_dims=...) (model.py:304,380-389)

    @property
    def grid(self) -> int:
        return self.img_size // self.patch

    @property
    def tokens(self) -> int:
        return 1 + self.num_frames * self.grid * self.grid

    @property
    def head_dims(self) -> List[int]:
        # model.py:380-383
        if self.embed_dims is not None:
            return list(self.embed_dims)
        return [(self.embed_dim * self.num_frames) // (2**i) for i in range(5)]


_VARIANTS = {
    # model.py:128-168 (embed_dim, depth, heads, patch)
    "prithvi_eo_tiny": (256, 4, 4, 16),
    "prithvi_eo_v1_100": (768, 12, 12, 16),
    "prithvi_eo_v2_100": (768, 12, 12, 16),
    "prithvi_eo_v2_300": (1024, 24, 16, 16),
    # model.py:147-153: coords_encoding time + location, coords_scale_learn=True.  The encoders (pritvhi.py:273-367) are built but
    # PrithviViT.forward (pritvhi.py:498-530) never calls them: same arithmetic as prithvi_eo_v2_300, two extra (1,) parameters.
    "prithvi_eo_v2_300_tl": (1024, 24, 16, 16),
    # model.py:154-167: 16 heads of 80, patch 14 (16 x 16 tokens of a 224 chip); head kernels [5, 5, 5, 7] (model.py:169-177)
    "prithvi_eo_v2_600": (1280, 32, 16, 14),
    "prithvi_eo_v2_600_tl": (1280, 32, 16, 14),
}
_HEAD_KERNELS = {"prithvi_eo_v2_600": (5, 5, 5, 7), "prithvi_eo_v2_600_tl": (5, 5, 5, 7)}  # model.py:169-177; (3, 3, 3, 3) otherwise


def make_config(
    variant: str = "prithvi_eo_v1_100",
    temporal_step: int = 1,
    num_classes: int = 2,
    image_size: int = 224,
    depth: int = -1,
) -> OracleConfig:
    d, l, h, p = _VARIANTS[variant]
    if depth != -1:  # model.py:208-209
        l = depth
    return OracleConfig(
        variant=variant,
        embed_dim=d,
        depth=l,
        num_heads=h,
        patch=p,
        num_frames=temporal_step,
        img_size=image_size,
        num_classes=num_classes,
        head_kernels=_HEAD_KERNELS.get(variant, (3, 3, 3, 3)),
    )


# --------------------------------------------------------------------------------------
# Positional embedding (instageo/model/pritvhi.py:67-127)
# --------------------------------------------------------------------------------------


def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """pritvhi.py:67-89.  omega is float32; the outer product promotes to float64."""
    assert embed_dim % 2 == 0
    omega = np.arange(embed_dim // 2, dtype=np.float32)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000**omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_pos_embed_3d(embed_dim: int, grid: Tuple[int, int, int], cls_token: bool = True) -> np.ndarray:
    """pritvhi.py:92-127.  Column blocks are (w | h | t); the cls row is zeros."""
    assert embed_dim % 16 == 0
    t, h, w = grid
    wd = embed_dim // 16 * 6
    hd = embed_dim // 16 * 6
    td = embed_dim // 16 * 4
    we = sincos_1d(wd, np.arange(w))
    he = sincos_1d(hd, np.arange(h))
    te = sincos_1d(td, np.arange(t))
    we = np.tile(we, (t * h, 1))
    he = np.tile(np.repeat(he, w, axis=0), (t, 1))
    te = np.repeat(te, h * w, axis=0)
    pe = np.concatenate((we, he, te), axis=1)
    if cls_token:
        pe = np.concatenate([np.zeros([1, embed_dim]), pe], axis=0)
    return pe


# --------------------------------------------------------------------------------------
# State-dict key contract and a seeded weight recipe
# --------------------------------------------------------------------------------------


def state_dict_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    """Keys/shapes of reference ``PrithviSeg.state_dict()`` (model.py:344,386-390)."""
    d, hid = cfg.embed_dim, cfg.embed_dim * cfg.mlp_ratio
    s: Dict[str, Tuple[int, ...]] = {}
    e = "prithvi_encoder."
    s[e + "cls_token"] = (1, 1, d)
    s[e + "pos_embed"] = (1, cfg.tokens, d)
    s[e + "patch_embed.proj.weight"] = (d, cfg.in_chans, 1, cfg.patch, cfg.patch)
    s[e + "patch_embed.proj.bias"] = (d,)
    if cfg.variant.endswith("_tl"):  # pritvhi.py:431-437, registered after patch_embed and before the blocks
        s[e + "temporal_embed_enc.scale"] = (1,)
        s[e + "location_embed_enc.scale"] = (1,)
    for i in range(cfg.depth):
        b = f"{e}blocks.{i}."
        s[b + "norm1.weight"] = (d,)
        s[b + "norm1.bias"] = (d,)
        s[b + "attn.qkv.weight"] = (3 * d, d)
        s[b + "attn.qkv.bias"] = (3 * d,)
        s[b + "attn.proj.weight"] = (d, d)
        s[b + "attn.proj.bias"] = (d,)
        s[b + "norm2.weight"] = (d,)
        s[b + "norm2.bias"] = (d,)
        s[b + "mlp.fc1.weight"] = (hid, d)
        s[b + "mlp.fc1.bias"] = (hid,)
        s[b + "mlp.fc2.weight"] = (d, hid)
        s[b + "mlp.fc2.bias"] = (d,)
    s[e + "norm.weight"] = (d,)
    s[e + "norm.bias"] = (d,)
    dims = cfg.head_dims
    h = "segmentation_head."
    for i in range(4):
        k = cfg.head_kernels[i]
        s[f"{h}{i}.0.weight"] = (dims[i], dims[i + 1], 3, 3)  # ConvTranspose2d (Cin,Cout,3,3)
        s[f"{h}{i}.0.bias"] = (dims[i + 1],)
        s[f"{h}{i}.2.weight"] = (dims[i + 1], dims[i + 1], k, k)
        s[f"{h}{i}.2.bias"] = (dims[i + 1],)
        s[f"{h}{i}.3.weight"] = (dims[i + 1],)
        s[f"{h}{i}.3.bias"] = (dims[i + 1],)
        s[f"{h}{i}.3.running_mean"] = (dims[i + 1],)
        s[f"{h}{i}.3.running_var"] = (dims[i + 1],)
        s[f"{h}{i}.3.num_batches_tracked"] = ()
    s[h + "5.weight"] = (cfg.num_classes, dims[4], 1, 1)
    s[h + "5.bias"] = (cfg.num_classes,)
    return s


def make_state_dict(cfg: OracleConfig, seed: int = 1042, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded weights by *recipe* (numpy PCG64, key order of ``state_dict_shapes``).

    Not the reference's initialiser (pritvhi.py:463-477 uses torch's RNG): a recipe both
    sides can regenerate so that fixtures need not store weights.  Scales are chosen so
    every stage has O(1) activations and non-trivial biases/affines: weights
    N(0, 2/(fan_in+fan_out)), biases N(0, 0.02), norm weights 1+N(0,0.1), running_var
    U(0.5,1.5), running_mean N(0,0.1); ``pos_embed`` is the fixed sin-cos table.
    """
    rng = np.random.default_rng(seed)
    out: Dict[str, torch.Tensor] = {}
    for k, shp in state_dict_shapes(cfg).items():
        if k.endswith("pos_embed"):
            g = (cfg.num_frames, cfg.grid, cfg.grid)
            v = sincos_pos_embed_3d(cfg.embed_dim, g, True)[None].astype(np.float32)
        elif k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.int64)
            continue
        elif k.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, size=shp)
        elif k.endswith("running_mean"):
            v = rng.normal(0.0, 0.1, size=shp)
        elif k.endswith("cls_token"):
            v = rng.normal(0.0, 0.02, size=shp)
        elif len(shp) == 1 and (".norm" in k or k.endswith(".3.weight") or "norm.weight" in k) and k.endswith("weight"):
            v = 1.0 + rng.normal(0.0, 0.1, size=shp)
        elif len(shp) == 1:
            v = rng.normal(0.0, 0.02, size=shp)
        else:
            if ".0.weight" in k and "segmentation_head" in k:  # ConvTranspose (Cin,Cout,kh,kw)
                fan_in, fan_out = shp[0] * 9 / 4.0, shp[1] * 9 / 4.0
            else:
                rf = int(np.prod(shp[2:])) if len(shp) > 2 else 1
                fan_in, fan_out = shp[1] * rf, shp[0] * rf
            v = rng.normal(0.0, math.sqrt(2.0 / (fan_in + fan_out)), size=shp)
        out[k] = torch.from_numpy(np.asarray(v, dtype=np.float64)).to(dtype)
    return out


# --------------------------------------------------------------------------------------
# Network forward (pritvhi.py:248-270,498-530 ; timm Block ; model.py:392-419)
# --------------------------------------------------------------------------------------


def patch_embed(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """pritvhi.py:243-245,266-268: Conv3d(k=s=(1,p,p)) then flatten(2).transpose(1,2).

    Restated as an explicit patch gather + matmul; token order (t, row, col).
    """
    B, C, T, H, W = x.shape
    p = w.shape[-1]
    gh, gw = H // p, W // p
    xp = x[:, :, :, : gh * p, : gw * p].reshape(B, C, T, gh, p, gw, p)
    xp = xp.permute(0, 2, 3, 5, 1, 4, 6).reshape(B, T * gh * gw, C * p * p)
    return xp @ w.reshape(w.shape[0], -1).t() + b


def vit_block(x: torch.Tensor, sd: Dict[str, torch.Tensor], pre: str, heads: int) -> torch.Tensor:
    """timm 1.0.20 ``Block.forward`` as configured at pritvhi.py:446-456 (third-party)."""
    B, N, D = x.shape
    hd = D // heads
    h = F.layer_norm(x, (D,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], 1e-5)
    qkv = F.linear(h, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q * hd**-0.5) @ k.transpose(-2, -1)
    att = att.softmax(dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, N, D)
    x = x + F.linear(o, sd[pre + "attn.proj.weight"], sd[pre + "attn.proj.bias"])
    h = F.layer_norm(x, (D,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], 1e-5)
    h = F.linear(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])
    h = F.gelu(h)  # nn.GELU() default = exact erf
    x = x + F.linear(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    return x


def interpolate_pos_encoding(cfg: OracleConfig, pos_embed: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """pritvhi.py:149-203 for an unchanged number of frames: the table itself when the input matches the configured grid,
    otherwise the patch rows resampled to (H/p, W/p) with bicubic interpolation, align_corners=True; the cls row is kept."""
    g = cfg.grid
    hp, wp = H // cfg.patch, W // cfg.patch
    if (hp, wp) == (g, g):
        return pos_embed
    cls_pe, patch_pe = pos_embed[:, :1], pos_embed[:, 1:]
    patch_pe = patch_pe.reshape(cfg.num_frames, g, g, cfg.embed_dim).permute(0, 3, 1, 2)
    patch_pe = F.interpolate(patch_pe, size=(hp, wp), mode="bicubic", align_corners=True)
    patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(1, -1, cfg.embed_dim)
    return torch.cat((cls_pe, patch_pe), dim=1)


def encoder_forward(
    cfg: OracleConfig, sd: Dict[str, torch.Tensor], img: torch.Tensor, stages: Optional[dict] = None
) -> torch.Tensor:
    """pritvhi.py:498-530."""
    e = "prithvi_encoder."
    if img.dim() == 4 and cfg.num_frames == 1:  # pritvhi.py:507-509
        img = img.unsqueeze(2)
    x = patch_embed(img, sd[e + "patch_embed.proj.weight"], sd[e + "patch_embed.proj.bias"])
    if stages is not None:
        stages["patch_embed"] = x
    pos = interpolate_pos_encoding(cfg, sd[e + "pos_embed"], img.shape[-2], img.shape[-1])
    x = x + pos[:, 1:, :]
    cls = (sd[e + "cls_token"] + pos[:, :1, :]).expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x), dim=1)
    for i in range(cfg.depth):
        x = vit_block(x, sd, f"{e}blocks.{i}.", cfg.num_heads)
        if stages is not None and i == 0:
            stages["block0"] = x
    x = F.layer_norm(x, (cfg.embed_dim,), sd[e + "norm.weight"], sd[e + "norm.bias"], 1e-5)
    if stages is not None:
        stages["encoder_out"] = x
    return x


def features_to_image(cfg: OracleConfig, feats: torch.Tensor) -> torch.Tensor:
    """model.py:405-413: drop cls, permute(0,2,1), reshape(B,-1,g,g) => channel = d*T + t."""
    r = feats[:, 1:, :]
    side = int(np.sqrt(r.shape[1] // cfg.num_frames))
    return r.permute(0, 2, 1).reshape(feats.shape[0], -1, side, side)


def head_forward(
    cfg: OracleConfig,
    sd: Dict[str, torch.Tensor],
    x: torch.Tensor,
    training: bool = False,
    bn_momentum_update: Optional[dict] = None,
    stages: Optional[dict] = None,
) -> torch.Tensor:
    """model.py:349-390 with Dropout as identity (p forced to 0 / eval mode).

    ``training=True`` uses batch statistics in BatchNorm (nn.BatchNorm2d train mode,
    eps 1e-5, momentum 0.1); when ``bn_momentum_update`` is a dict the new running
    statistics are written into it.
    """
    h = "segmentation_head."
    for i in range(4):
        x = F.conv_transpose2d(x, sd[f"{h}{i}.0.weight"], sd[f"{h}{i}.0.bias"], stride=2, padding=1, output_padding=1)
        x = F.conv2d(x, sd[f"{h}{i}.2.weight"], sd[f"{h}{i}.2.bias"], padding=1)
        rm, rv = sd[f"{h}{i}.3.running_mean"], sd[f"{h}{i}.3.running_var"]
        if training:
            if bn_momentum_update is not None:
                n = x.numel() / x.shape[1]
                mean = x.mean(dim=(0, 2, 3))
                var_u = x.var(dim=(0, 2, 3), unbiased=False) * (n / (n - 1))
                bn_momentum_update[f"{h}{i}.3.running_mean"] = (0.9 * rm + 0.1 * mean).detach()
                bn_momentum_update[f"{h}{i}.3.running_var"] = (0.9 * rv + 0.1 * var_u).detach()
            x = F.batch_norm(x, None, None, sd[f"{h}{i}.3.weight"], sd[f"{h}{i}.3.bias"], True, 0.1, 1e-5)
        else:
            x = F.batch_norm(x, rm, rv, sd[f"{h}{i}.3.weight"], sd[f"{h}{i}.3.bias"], False, 0.1, 1e-5)
        x = F.relu(x)
        if stages is not None:
            stages[f"head{i}"] = x
    return F.conv2d(x, sd[h + "5.weight"], sd[h + "5.bias"])


def prithvi_seg_forward(
    cfg: OracleConfig,
    sd: Dict[str, torch.Tensor],
    img: torch.Tensor,
    training: bool = False,
    stages: Optional[dict] = None,
    bn_momentum_update: Optional[dict] = None,
) -> torch.Tensor:
    """``PrithviSeg.forward`` (model.py:392-419), dropout disabled."""
    feats = encoder_forward(cfg, sd, img, stages)
    fimg = features_to_image(cfg, feats)
    if stages is not None:
        stages["features"] = fimg
    return head_forward(cfg, sd, fimg, training, bn_momentum_update, stages)


# --------------------------------------------------------------------------------------
# Loss / predictions (instageo/model/segmentation.py:84-87,107-147)
# --------------------------------------------------------------------------------------


def seg_loss(
    logits: torch.Tensor, labels: torch.Tensor, class_weights: Optional[torch.Tensor], ignore_index: int
) -> torch.Tensor:
    """segmentation.py:85-87,117-122: CE(weight, ignore_index, 'none') then loss[mask].mean()
    i.e. sum(w_y * nll) / (#valid pixels) -- NOT torch's weighted mean."""
    labels = labels.long()
    mask = labels.ne(ignore_index)
    loss = F.cross_entropy(logits, labels, weight=class_weights, ignore_index=ignore_index, reduction="none")
    return loss[mask].mean()


def seg_predictions(logits: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """segmentation.py:125-126."""
    return torch.argmax(logits, dim=1), torch.softmax(logits.detach(), dim=1)


# --------------------------------------------------------------------------------------
# Streaming confusion matrix (instageo/model/metrics.py:50-166)
# --------------------------------------------------------------------------------------


def _safe_div(num: np.ndarray, den: np.ndarray) -> np.ndarray:
    """metrics.py:50-55."""
    den = den.astype(float)
    out = np.zeros_like(den, dtype=float)
    np.divide(num, den, out=out, where=den != 0)
    return out


def confusion_matrix(y_true: np.ndarray, y_pred: np.ndarray, k: int, ignore_index: Optional[int]) -> np.ndarray:
    """metrics.py:86-108: int64 bincount of y_true*k + y_pred over non-ignored pixels."""
    y_true = np.asarray(y_true).ravel().astype(np.int64)
    y_pred = np.asarray(y_pred).ravel().astype(np.int64)
    if ignore_index is not None:
        m = y_true != ignore_index
        y_true, y_pred = y_true[m], y_pred[m]
    if y_true.size == 0:
        return np.zeros((k, k), dtype=np.int64)
    return np.bincount(y_true * k + y_pred, minlength=k * k).reshape(k, k).astype(np.int64)


def confusion_metrics(matrix: np.ndarray) -> dict:
    """metrics.py:110-166 (accuracy, macro precision/recall/f1/jaccard + per-class)."""
    tp = np.diag(matrix)
    fp = matrix.sum(axis=0) - tp
    fn = matrix.sum(axis=1) - tp
    total = matrix.sum()
    prec = _safe_div(tp, tp + fp)
    rec = _safe_div(tp, tp + fn)
    f1 = _safe_div(2 * prec * rec, prec + rec)
    jac = _safe_div(tp, tp + fp + fn)
    return {
        "accuracy": float("nan") if total == 0 else tp.sum() / total,
        "precision": prec.mean(),
        "recall": rec.mean(),
        "f1": f1.mean(),
        "jaccard": jac.mean(),
        "precision_per_class": prec.tolist(),
        "recall_per_class": rec.tolist(),
        "f1_per_class": f1.tolist(),
        "jaccard_per_class": jac.tolist(),
    }


# --------------------------------------------------------------------------------------
# Dataset side: normalise / layout / window tiling (instageo/model/dataloader.py)
# --------------------------------------------------------------------------------------


def distillation_loss(student_logits: torch.Tensor, teacher_logits: torch.Tensor, labels: torch.Tensor, ignore_index: int,
                      class_weights: Optional[torch.Tensor] = None):
    """segmentation.py:352-378 (_compute_loss): ce = CrossEntropyLoss(weight, ignore_index, 'none')[valid].mean();
    distill = KLDivLoss('batchmean')(log_softmax(student[valid]), softmax(teacher[valid])) = sum / #valid;
    returns (total, ce, distill)."""
    k = student_logits.shape[1]
    lab = labels.long()
    ce = F.cross_entropy(student_logits, lab, weight=class_weights, ignore_index=ignore_index, reduction="none")
    valid = lab.ne(ignore_index).reshape(-1)
    s = student_logits.permute(0, 2, 3, 1).reshape(-1, k)[valid]
    t = teacher_logits.permute(0, 2, 3, 1).reshape(-1, k)[valid]
    ce = ce.reshape(-1)[valid].mean()
    distill = F.kl_div(F.log_softmax(s, dim=1), F.softmax(t, dim=1), reduction="batchmean")
    return ce + distill, ce, distill


def regression_loss(outputs: torch.Tensor, labels: torch.Tensor, ignore_index: float, use_log_scale: bool = False):
    """regression.py:153-168: outputs (B,1,H,W).squeeze(1), mask = labels != ignore_index, labels -> log1p when
    use_log_scale, MSE mean over the masked pixels; returns (loss, de-scaled predictions, de-scaled labels)."""
    out = outputs.squeeze(1)
    mask = labels.ne(ignore_index)
    lab = torch.log1p(labels) if use_log_scale else labels
    o, l = out[mask], lab[mask]
    loss = ((o - l) ** 2).mean()
    preds, l2 = o.detach(), l
    if use_log_scale:
        preds, l2 = torch.expm1(preds), torch.expm1(l2)
    return loss, preds, l2


def regression_distillation_loss(student: torch.Tensor, teacher: torch.Tensor, labels: torch.Tensor, ignore_index: float,
                                 use_log_scale: bool = False):
    """regression.py:477-534 (PrithviDistillationRegressionModule._shared_step + _compute_loss): mask = labels != ignore_index;
    labels and TEACHER outputs go through log1p under use_log_scale; total = mean((s - y')^2) + mean((s - t')^2) over the mask.
    Returns (total, mse, distill)."""
    mask = labels.ne(ignore_index)
    lab = torch.log1p(labels) if use_log_scale else labels
    t = torch.log1p(teacher) if use_log_scale else teacher
    s_, t_, l_ = student.squeeze(1)[mask], t.squeeze(1)[mask], lab[mask]
    mse = ((s_ - l_) ** 2).mean()
    kd = ((s_ - t_) ** 2).mean()
    return mse + kd, mse, kd


def regression_sums(y_true: np.ndarray, y_pred: np.ndarray, ee_bias: float = 0.05, ee_coef: float = 0.15) -> List[float]:
    """metrics.py:330-352 (RunningRegressionMetrics.update) as one batch, accumulated in float64:
    [n, Sx, Sy, Sxy, Sxx, Syy, S|e|, See, #(|e| <= ee_bias + ee_coef x)]."""
    x = np.asarray(y_true, dtype=np.float64).ravel()
    y = np.asarray(y_pred, dtype=np.float64).ravel()
    e = np.abs(y - x)
    return [float(x.size), x.sum(), y.sum(), (x * y).sum(), (x * x).sum(), (y * y).sum(), e.sum(), (e * e).sum(),
            float(np.sum(e <= ee_bias + ee_coef * x))]


def regression_metrics(sums: List[float], include_ee: bool = False) -> dict:
    """metrics.py:354-420: mae, rmse, r2 = 1 - SSE / (Sxx - n xm^2), pearson, expected-error percentage."""
    n, sx, sy, sxy, sxx, syy, sae, sse, nee = sums
    nan = float("nan")
    out = {"mae": sae / n if n else nan, "rmse": float(np.sqrt(sse / n)) if n else nan, "r2_score": nan, "pearson_corrcoef": nan,
           "ee_percentage": (nee / n * 100 if n else nan) if include_ee else None}
    if n >= 2:
        xm, ym = sx / n, sy / n
        ss_tot = sxx - n * xm * xm
        if ss_tot != 0:
            out["r2_score"] = 1 - sse / ss_tot
        std_x, std_y = np.sqrt(sxx - n * xm * xm), np.sqrt(syy - n * ym * ym)
        if std_x != 0 and std_y != 0:
            out["pearson_corrcoef"] = float((sxy - n * xm * ym) / (std_x * std_y))
    return out


def auc_histograms(y_true: np.ndarray, y_score: np.ndarray, num_classes: int, n_bins: int = 1024, min_score: float = 0.0,
                   max_score: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    """metrics.py:204-236 (RunningAUC._bin / update): per class c, histogram of the class-c score of the positives
    (y == c) and of the negatives; bin = int((clamp(s) - min) / (max - min) * (n_bins - 1)) in the scores' own dtype."""
    y_true = np.asarray(y_true).ravel()
    y_score = np.asarray(y_score)
    pos = np.zeros((num_classes, n_bins), dtype=np.int64)
    neg = np.zeros((num_classes, n_bins), dtype=np.int64)
    for c in range(num_classes):
        s = np.minimum(max_score, np.maximum(min_score, y_score[:, c]))
        bins = ((s - min_score) / (max_score - min_score) * (n_bins - 1)).astype(np.int64)
        np.add.at(pos[c], bins[y_true == c], 1)
        np.add.at(neg[c], bins[y_true != c], 1)
    return pos, neg


def auc_score(pos: np.ndarray, neg: np.ndarray) -> Tuple[float, List[float]]:
    """metrics.py:238-264: AUC_c = sum_bins pos * (negatives in lower bins + half of the negatives in the same bin) /
    (n_pos n_neg), NaN without positives or negatives; macro = nanmean."""
    per = []
    for c in range(pos.shape[0]):
        n_pos, n_neg = int(pos[c].sum()), int(neg[c].sum())
        if n_pos == 0 or n_neg == 0:
            per.append(float("nan"))
            continue
        auc, cum = 0.0, 0
        for p_, n_ in zip(pos[c].tolist(), neg[c].tolist()):
            auc += p_ * cum + 0.5 * p_ * n_
            cum += n_
        per.append(auc / (n_pos * n_neg))
    return float(np.nanmean(np.array(per))), per


def normalize_chip(chip: np.ndarray, mean: List[float], std: List[float], temporal_size: int) -> np.ndarray:
    """dataloader.py:495-524: (T*C,H,W) -> reshape (T,C,H,W) -> (x-mean_c)/std_c -> (C,T,H,W) f32.

    ToTensor on PIL mode-F images does not rescale, so this is the whole arithmetic;
    torchvision Normalize computes ``(x - mean) / std`` in float32.
    """
    x = torch.from_numpy(np.asarray(chip)).float()
    tc, h, w = x.shape
    x = x.reshape(temporal_size, -1, h, w)
    m = torch.tensor(mean, dtype=torch.float32).view(1, -1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, -1, 1, 1)
    x = (x - m) / s
    return x.permute(1, 0, 2, 3).contiguous().numpy()


def crop_flip_chip(chip: np.ndarray, label: Optional[np.ndarray], top: int, left: int, hflip: bool, vflip: bool,
                   im_size: int) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """dataloader.py:58-77 (crop at (i=top, j=left) to im_size) followed by :80-141 (hflip = reverse the last axis,
    vflip = reverse the row axis), applied to every band and to the label alike.  The random draws themselves
    (RandomCrop.get_params, random.random() < p) stay on the host; this is the data movement they select."""
    x = np.asarray(chip)[:, top : top + im_size, left : left + im_size]
    y = None if label is None else np.asarray(label)[top : top + im_size, left : left + im_size]
    if hflip:
        x = x[:, :, ::-1]
        y = None if y is None else y[:, ::-1]
    if vflip:
        x = x[:, ::-1, :]
        y = None if y is None else y[::-1, :]
    return np.ascontiguousarray(x), (None if y is None else np.ascontiguousarray(y))


def compute_class_weights(counts: Dict[int, int]) -> List[float]:
    """pipeline_utils.py:183-203: w_c = total / (number of classes present * count_c), list indexed by class id."""
    total = sum(counts.values())
    ncls = len(counts)
    out = [0.0] * (int(max(counts.keys())) + 1)
    for cls, cnt in counts.items():
        out[int(cls)] = total / (ncls * cnt)
    return out


def compute_stats(batches, is_reg_task: bool = False):
    """pipeline_utils.py:207-254 (mode=stats).  ``batches`` yields (data (B,C,T,H,W), label (B,H,W)).

    mean_c = (1/N) sum_b mean_{t,h,w} x ; std_c = sqrt((1/N) sum_b biased-var_{t,h,w} x) -- the average of the per-chip
    variances, not the pooled variance; class weights from the label counts with the ignore value -1 removed.
    Accumulated in float64 here (the reference accumulates float32 tensors)."""
    mean = 0.0
    var = 0.0
    n = 0
    counts: Dict[int, int] = {}
    for data, label in batches:
        d = torch.as_tensor(np.asarray(data)).double()
        b = d.shape[0]
        d = d.reshape(b, d.shape[1], -1)
        n += b
        mean = mean + d.mean(2).sum(0)
        var = var + d.var(2, unbiased=False).sum(0)
        if not is_reg_task:
            vals, cnts = np.unique(np.asarray(label), return_counts=True)
            for v, c in zip(vals, cnts):
                counts[v] = counts.get(v, 0) + int(c)
    mean = mean / n
    std = torch.sqrt(var / n)
    weights = None
    if not is_reg_task:
        counts.pop(-1, None)
        weights = compute_class_weights(counts)
    return mean.tolist(), std.tolist(), weights


def window_origins(img_size: int, crop_size: int, stride: int) -> List[Tuple[int, int]]:
    """dataloader.py:655-664: ``for top in range(0,S-crop+1,stride) for left in ...`` -> (top,left)."""
    return [
        (top, left)
        for top in range(0, img_size - crop_size + 1, stride)
        for left in range(0, img_size - crop_size + 1, stride)
    ]


# --------------------------------------------------------------------------------------
# Optimiser / scheduler (instageo/model/base.py:115-133)
# --------------------------------------------------------------------------------------


def adamw_step(
    p: torch.Tensor,
    g: torch.Tensor,
    m: torch.Tensor,
    v: torch.Tensor,
    step: int,
    lr: float,
    wd: float = 1e-2,
    b1: float = 0.9,
    b2: float = 0.999,
    eps: float = 1e-8,
) -> None:
    """torch.optim.AdamW single-tensor update (base.py:124-126 uses torch defaults), in place."""
    p.mul_(1 - lr * wd)
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1**step
    bc2 = 1 - b2**step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def cosine_warm_restarts_lr(base_lr: float, epoch: int, t0: int = 10, t_mult: int = 2, eta_min: float = 0.0) -> float:
    """CosineAnnealingWarmRestarts(T_0=10,T_mult=2,eta_min=0) stepped per epoch (base.py:128-131)."""
    t_i, t_cur = t0, epoch
    while t_cur >= t_i:
        t_cur -= t_i
        t_i *= t_mult
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t_cur / t_i)) / 2


# --------------------------------------------------------------------------------------
# A full reference-semantics training step on CPU (used for parity + cpu_baseline)
# --------------------------------------------------------------------------------------


def train_step_reference(
    cfg: OracleConfig,
    sd: Dict[str, torch.Tensor],
    img: torch.Tensor,
    labels: torch.Tensor,
    class_weights: Optional[torch.Tensor],
    ignore_index: int,
    wanted_grads: Optional[List[str]] = None,
) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, torch.Tensor]]:
    """forward (BN in train mode, dropout p=0) + loss + autograd backward.

    Returns (logits, loss, grads) where grads maps state-dict keys to d loss / d param.
    """
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k and not k.endswith("pos_embed")]
    leaf = {k: sd[k].detach().clone().requires_grad_(True) for k in names}
    full = dict(sd)
    full.update(leaf)
    logits = prithvi_seg_forward(cfg, full, img, training=True)
    loss = seg_loss(logits, labels, class_weights, ignore_index)
    keys = wanted_grads if wanted_grads is not None else names
    gs = torch.autograd.grad(loss, [leaf[k] for k in keys])
    return logits.detach(), loss.detach(), dict(zip(keys, gs))
