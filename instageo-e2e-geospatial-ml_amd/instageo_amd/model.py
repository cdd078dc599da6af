"""``PrithviSeg`` for MI355X: same constructor / ``forward`` / ``state_dict`` contract as the reference
(``instageo/model/model.py:292-419``, encoder ``instageo/model/pritvhi.py:370-530``), every arithmetic op
executed by the HIP library through :mod:`instageo_amd.ops`.

Design
------
* All parameters live in ONE flat fp32 buffer (plus flat grad / bf16-shadow buffers): AdamW, gradient
  all-reduce buckets and the bf16 operand refresh are single streaming passes; ``nn.Parameter`` objects are
  views into it so ``state_dict()`` / ``load_state_dict(strict=True)`` keep the reference's keys and shapes.
  Conv weights are *stored* ``[Cout][9][Cin]`` (the layout the gather-GEMM wants) and exposed as permuted
  views with the PyTorch shapes.
* ``SegEngine`` runs the explicit forward / backward schedule on preallocated workspaces (no autograd
  graph, graph-capturable).  ``PrithviSeg.forward`` wraps it in one ``torch.autograd.Function`` so that
  ``loss.backward()`` of an unmodified training loop still works.
* precision: ``"bf16"`` (bf16 MFMA operands, fp32 accumulate/residual stream) or ``"bf16x3"`` (split
  hi/lo operands, fp32-grade results; used for the 1e-3 parity tests).
"""
from __future__ import annotations

import itertools
import math
import os
import weakref
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .ops import BT

# --------------------------------------------------------------------------------------------------
# configuration (reference: model.py:39-177)
# --------------------------------------------------------------------------------------------------
PRITHVI_VARIANTS = {
    # variant: (embed_dim, depth, num_heads, patch, default num_frames)
    "prithvi_eo_tiny": (256, 4, 4, 16, 1),
    "prithvi_eo_v1_100": (768, 12, 12, 16, 3),
    "prithvi_eo_v2_100": (768, 12, 12, 16, 4),
    "prithvi_eo_v2_300": (1024, 24, 16, 16, 4),
    # coords_encoding=["time", "location"], coords_scale_learn=True (model.py:147-153): the reference builds a TemporalEncoder and a
    # LocationEncoder (pritvhi.py:273-367) whose only state is a trainable ``scale`` (1,) = 0.1 each -- and never calls them:
    # PrithviViT.forward (pritvhi.py:498-530) takes no coordinates.  The variant therefore computes exactly what prithvi_eo_v2_300
    # computes and carries two extra state_dict entries that receive no gradient.
    "prithvi_eo_v2_300_tl": (1024, 24, 16, 16, 4),
    # model.py:154-167: 16 heads of 80 (attention_g.hip), patch 14 (16 x 16 tokens of a 224 chip), and -- model.py:169-177 -- decode-head
    # convolutions of kernel size [5, 5, 5, 7] with padding 1, which shrink every upsampled map (32 -> 30, 60 -> 58, 116 -> 114, 228 -> 224)
    "prithvi_eo_v2_600": (1280, 32, 16, 14, 4),
    "prithvi_eo_v2_600_tl": (1280, 32, 16, 14, 4),
}
HEAD_KERNELS = {k: (3, 3, 3, 3) for k in PRITHVI_VARIANTS}
HEAD_KERNELS.update({"prithvi_eo_v2_600": (5, 5, 5, 7), "prithvi_eo_v2_600_tl": (5, 5, 5, 7)})
TL_VARIANTS = ("prithvi_eo_v2_300_tl", "prithvi_eo_v2_600_tl")


@dataclass
class SegConfig:
    variant: str = "prithvi_eo_v1_100"
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    patch: int = 16
    in_chans: int = 6
    num_frames: int = 1
    img_size: int = 224
    num_classes: int = 2
    mlp_ratio: int = 4
    drop_p: float = 0.1  # nn.Dropout(0.1) x5 in the head (model.py:369,388)
    head_kernels: Tuple[int, int, int, int] = (3, 3, 3, 3)  # nn.Conv2d kernel sizes of the four upscaling blocks (model.py:169-177)
    embed_dims: Optional[Tuple[int, int, int, int, int]] = None  # custom decode-head widths (PrithviSeg(embed_dims=...), model.py:304,380-389)

    @property
    def grid(self) -> int:
        return self.img_size // self.patch

    @property
    def G(self) -> int:
        return self.grid * self.grid

    @property
    def tokens(self) -> int:
        return 1 + self.num_frames * self.G

    @property
    def head_dims(self) -> List[int]:
        if self.embed_dims is not None:
            return list(self.embed_dims)
        return [(self.embed_dim * self.num_frames) // (2**i) for i in range(5)]  # model.py:380-383

    @property
    def patch_k(self) -> int:
        return self.in_chans * self.patch * self.patch

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def head_sizes(self) -> List[Tuple[int, int, int]]:
        """Per upscaling block (input side, ConvTranspose output side = 2 x, Conv2d(k, padding=1) output side = 2 x + 3 - k):
        (14, 28, 28) ... (112, 224, 224) for 3 x 3 kernels; (16, 32, 30), (30, 60, 58), (58, 116, 114), (114, 228, 224) for the 600M head."""
        out, h = [], self.grid
        for k in self.head_kernels:
            out.append((h, 2 * h, 2 * h + 3 - k))
            h = 2 * h + 3 - k
        return out

    @property
    def out_size(self) -> int:
        return self.head_sizes[-1][2]


def make_seg_config(variant: str, temporal_step: int, image_size: int, num_classes: int, depth: int = -1,
                    in_chans: int = 6, embed_dims: Optional[List[int]] = None) -> SegConfig:
    if variant not in PRITHVI_VARIANTS:
        raise KeyError(f"unknown Prithvi variant {variant!r}")
    d, l, h, p, _ = PRITHVI_VARIANTS[variant]
    if depth != -1:  # model.py:208-209
        l = depth
    assert image_size % p == 0, "image_size must be divisible by the patch size"
    assert d % h == 0 and d // h in (64, 80), "head_dim must be 64 or 80 (attention2.hip / attention_g.hip)"
    if embed_dims is not None:
        # model.py:380-389: the first width is the ConvTranspose's input = the encoder's feature channels D * T; the kernels move
        # 16-byte (8-channel) units
        embed_dims = tuple(int(x) for x in embed_dims)
        if len(embed_dims) != 5 or embed_dims[0] != d * temporal_step or any(x <= 0 or x % 8 for x in embed_dims):
            raise ValueError(f"embed_dims must be five positive multiples of 8 starting with embed_dim * temporal_step = {d * temporal_step}, got {list(embed_dims)}")
    return SegConfig(variant, d, l, h, p, in_chans, temporal_step, image_size, num_classes, head_kernels=HEAD_KERNELS[variant],
                     embed_dims=embed_dims)


# --------------------------------------------------------------------------------------------------
# positional embedding (pritvhi.py:67-127), float64 numpy -> f32 buffer exactly like the reference
# --------------------------------------------------------------------------------------------------
def _sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    omega = np.arange(embed_dim // 2, dtype=np.float32)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000**omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_3d_sincos_pos_embed(embed_dim: int, grid_size: Tuple[int, int, int], cls_token: bool = False) -> np.ndarray:
    """Fixed 3-D sin-cos table; column blocks (w | h | t) of widths 6D/16, 6D/16, 4D/16."""
    assert embed_dim % 16 == 0
    t, h, w = grid_size
    wd = hd = embed_dim // 16 * 6
    td = embed_dim // 16 * 4
    we = np.tile(_sincos_1d(wd, np.arange(w)), (t * h, 1))
    he = np.tile(np.repeat(_sincos_1d(hd, np.arange(h)), w, axis=0), (t, 1))
    te = np.repeat(_sincos_1d(td, np.arange(t)), h * w, axis=0)
    pe = np.concatenate((we, he, te), axis=1)
    if cls_token:
        pe = np.concatenate([np.zeros([1, embed_dim]), pe], axis=0)
    return pe


get_1d_sincos_pos_embed_from_grid = _sincos_1d  # the reference's name (pritvhi.py:67-88)


def get_1d_sincos_embed_from_grid_torch(embed_dim: int, pos: torch.Tensor) -> torch.Tensor:
    """Torch twin of :func:`get_1d_sincos_pos_embed_from_grid` (pritvhi.py:130-146): (M,) positions -> (M, D), sin | cos."""
    assert embed_dim % 2 == 0
    omega = 1.0 / 10000 ** (torch.arange(embed_dim // 2, dtype=torch.float32, device=pos.device) / (embed_dim / 2.0))
    out = torch.einsum("m,d->md", pos.reshape(-1).float(), omega)
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)


def interpolate_pos_encoding(pos_embed: torch.Tensor, grid_size: Tuple[int, int, int], patch_size, shape: Tuple[int, int, int],
                             embed_dim: int) -> torch.Tensor:
    """Position table for an input of ``shape`` = (frames, height, width) given the table of token grid ``grid_size``
    (pritvhi.py:149-203): unchanged when the token grids agree; a changed frame count regenerates the sin-cos table for the new
    number of frames; the spatial grid is resampled per frame, bicubic with ``align_corners=True``; the cls row is kept."""
    tg, hg, wg = shape[0] // patch_size[0], shape[1] // patch_size[1], shape[2] // patch_size[2]
    if (tg, hg, wg) == tuple(grid_size):
        return pos_embed
    if tg != grid_size[0]:
        grid_size = (tg, grid_size[1], grid_size[2])
        pos_embed = torch.from_numpy(get_3d_sincos_pos_embed(pos_embed.shape[-1], grid_size, cls_token=True)).float().unsqueeze(0)
    cls_row, body = pos_embed[:, :1], pos_embed[:, 1:]
    body = body.reshape(*grid_size, embed_dim).permute(0, 3, 1, 2)
    body = torch.nn.functional.interpolate(body, size=(hg, wg), mode="bicubic", align_corners=True)
    return torch.cat((cls_row, body.permute(0, 2, 3, 1).reshape(1, -1, embed_dim)), dim=1)


# --------------------------------------------------------------------------------------------------
# flat parameter store
# --------------------------------------------------------------------------------------------------
@dataclass
class _Entry:
    name: str
    shape: Tuple[int, ...]  # API (PyTorch) shape
    offset: int
    numel: int
    kind: str  # "plain" | "conv" (Cout,Cin,k,k stored [Cout][k*k][Cin]) | "convT" (Cin,Cout,3,3 stored [Cout][9][Cin])

    def api_view(self, flat: torch.Tensor) -> torch.Tensor:
        seg = flat[self.offset : self.offset + self.numel]
        if self.kind == "plain":
            return seg.view(self.shape)
        if self.kind == "conv":
            co, ci, k = self.shape[0], self.shape[1], self.shape[2]
            return seg.view(co, k, k, ci).permute(0, 3, 1, 2)
        ci, co = self.shape[0], self.shape[1]
        return seg.view(co, 3, 3, ci).permute(3, 0, 1, 2)


def _param_specs(cfg: SegConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    d, hid = cfg.embed_dim, cfg.embed_dim * cfg.mlp_ratio
    e = "prithvi_encoder."
    s: List[Tuple[str, Tuple[int, ...], str]] = [
        (e + "cls_token", (1, 1, d), "plain"),
        (e + "patch_embed.proj.weight", (d, cfg.in_chans, 1, cfg.patch, cfg.patch), "plain"),
        (e + "patch_embed.proj.bias", (d,), "plain"),
    ]
    for i in range(cfg.depth):
        b = f"{e}blocks.{i}."
        s += [
            (b + "norm1.weight", (d,), "plain"), (b + "norm1.bias", (d,), "plain"),
            (b + "attn.qkv.weight", (3 * d, d), "plain"), (b + "attn.qkv.bias", (3 * d,), "plain"),
            (b + "attn.proj.weight", (d, d), "plain"), (b + "attn.proj.bias", (d,), "plain"),
            (b + "norm2.weight", (d,), "plain"), (b + "norm2.bias", (d,), "plain"),
            (b + "mlp.fc1.weight", (hid, d), "plain"), (b + "mlp.fc1.bias", (hid,), "plain"),
            (b + "mlp.fc2.weight", (d, hid), "plain"), (b + "mlp.fc2.bias", (d,), "plain"),
        ]  # fmt: skip
    s += [(e + "norm.weight", (d,), "plain"), (e + "norm.bias", (d,), "plain")]
    dims = cfg.head_dims
    h = "segmentation_head."
    for i in range(4):
        s += [
            (f"{h}{i}.0.weight", (dims[i], dims[i + 1], 3, 3), "convT"), (f"{h}{i}.0.bias", (dims[i + 1],), "plain"),
            (f"{h}{i}.2.weight", (dims[i + 1], dims[i + 1], cfg.head_kernels[i], cfg.head_kernels[i]), "conv"), (f"{h}{i}.2.bias", (dims[i + 1],), "plain"),
            (f"{h}{i}.3.weight", (dims[i + 1],), "plain"), (f"{h}{i}.3.bias", (dims[i + 1],), "plain"),
        ]  # fmt: skip
    s += [(h + "5.weight", (cfg.num_classes, dims[4], 1, 1), "plain"), (h + "5.bias", (cfg.num_classes,), "plain")]
    return s


class ParamStore:
    """Flat fp32 parameters (+grad, +bf16 shadow) with named, 8-element-aligned segments."""

    ALIGN = 8

    def __init__(self, cfg: SegConfig, device: torch.device):
        self.cfg = cfg
        self.entries: Dict[str, _Entry] = {}
        off = 0
        self.encoder_end = 0
        for name, shape, kind in _param_specs(cfg):
            n = int(np.prod(shape))
            self.entries[name] = _Entry(name, tuple(shape), off, n, kind)
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            if name.startswith("prithvi_encoder."):
                self.encoder_end = off
        self.total = off
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grad: Optional[torch.Tensor] = None
        self.shadow: Optional[BT] = None
        self.shadow_split: Optional[bool] = None
        # transposed operand copy of the Block linears' weights (dgrad as a K-contiguous GEMM): per block
        # [qkv^T (D,3D) | proj^T (D,D) | fc1^T (D,4D) | fc2^T (4D,D)] = 12 D^2 elements
        self.shadow_t: Optional[BT] = None
        d = cfg.embed_dim
        self.t_block = 12 * d * d
        self.t_offsets = {"attn.qkv.weight": (0, 3 * d, d), "attn.proj.weight": (3 * d * d, d, d),
                          "mlp.fc1.weight": (4 * d * d, 4 * d, d), "mlp.fc2.weight": (8 * d * d, d, 4 * d)}  # name: (offset, R, C)

    def seg(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        e = self.entries[name]
        return buf[e.offset : e.offset + e.numel]

    def ensure_grad(self) -> torch.Tensor:
        if self.grad is None or self.grad.device != self.flat.device:
            self.grad = torch.zeros_like(self.flat)
        return self.grad

    def refresh_shadow(self, split: bool) -> BT:
        """(Re)build the bf16 (hi/lo) operand copy of all parameters from the fp32 master."""
        if self.shadow is None or self.shadow_split != split or self.shadow.hi.device != self.flat.device:
            self.shadow = BT.empty((self.total,), split, self.flat.device)
            self.shadow_split = split
        ops.split_bf16(self.flat, self.shadow)
        return self.shadow

    def refresh_shadow_t(self) -> None:
        """Re-transpose the Block linears' bf16 (hi/lo) weights from the shadow: 4 batched launches (x2 in split mode)."""
        cfg, sh = self.cfg, self.shadow
        if cfg.depth == 0 or cfg.embed_dim % 64:
            return
        split = sh.lo is not None
        if self.shadow_t is None or self.shadow_t.split != split or self.shadow_t.hi.device != sh.hi.device:
            self.shadow_t = BT.empty((cfg.depth * self.t_block,), split, sh.hi.device)
        e0 = "prithvi_encoder.blocks.0."
        stride = self.entries["prithvi_encoder.blocks.1.norm1.weight"].offset - self.entries[e0 + "norm1.weight"].offset if cfg.depth > 1 else 0
        for name, (toff, R, C) in self.t_offsets.items():
            so = self.entries[e0 + name].offset
            src = BT(sh.hi[so:], None if sh.lo is None else sh.lo[so:])
            dst = BT(self.shadow_t.hi[toff:], None if self.shadow_t.lo is None else self.shadow_t.lo[toff:])
            ops.transpose_bf16(src, dst, R, C, cfg.depth, stride, self.t_block)

    def wt(self, block: int, name: str) -> Optional[BT]:
        """Transposed bf16 operand view (C, R) of a Block linear weight, or None when no transposed copy is kept."""
        if self.shadow_t is None:
            return None
        toff, R, C = self.t_offsets[name]
        o = block * self.t_block + toff
        return BT(self.shadow_t.hi[o : o + R * C], None if self.shadow_t.lo is None else self.shadow_t.lo[o : o + R * C])

    # data parallel: lowest flat offset whose deferred all-gather is still in flight (distributed.ShardedGradSync), None = none pending.
    # Reading an operand at or above it would race the gather: w() refuses (the engine's readers call SegEngine._need first, which waits).
    pending_from: Optional[Callable[[], Optional[int]]] = None

    def w(self, name: str) -> BT:
        """bf16 operand view of a parameter (storage layout)."""
        e = self.entries[name]
        if self.pending_from is not None:
            lo = self.pending_from()
            if lo is not None and e.offset + e.numel > lo:
                raise RuntimeError(f"ParamStore.w({name!r}): the all-gather of the operand copy from flat offset {lo} is still in flight; "
                                   "call the engine's param_wait / ShardedGradSync.wait_params() before reading it")
        sh = self.shadow
        return BT(sh.hi[e.offset : e.offset + e.numel], None if sh.lo is None else sh.lo[e.offset : e.offset + e.numel])


# --------------------------------------------------------------------------------------------------
# the explicit forward / backward engine
# --------------------------------------------------------------------------------------------------
class SegEngine:
    """Runs the Prithvi segmentation network on HIP kernels with preallocated workspaces."""

    def __init__(self, cfg: SegConfig, store: ParamStore, buffers: Dict[str, torch.Tensor], precision: str = "bf16"):
        assert precision in ("bf16", "bf16x3")
        self.cfg, self.store, self.buffers = cfg, store, buffers
        self.precision = precision
        self.split = precision == "bf16x3"
        self._ws: Dict[Any, Dict[str, Any]] = {}
        self.shadow_dirty = True
        self.shadow_t_dirty = True  # the transposed weight copy lags the shadow until the next backward needs it
        self.drop_seed = 1042
        self._drop_step: Optional[torch.Tensor] = None  # device uint32 counter mixed into the dropout hash each step
        self.freeze_backbone = False
        self.on_grad_ready: Optional[Callable[[int, int], None]] = None
        self.master_sync: Optional[Callable[[], None]] = None  # distributed.ShardedGradSync.gather_master when the fp32 masters are sharded
        self.master_complete: Optional[Callable[[], bool]] = None  # ... and whether they are complete on this rank right now
        # data parallel: wait for the deferred all-gather of the parameters below a flat offset (ShardedGradSync.wait_params)
        self.param_wait: Optional[Callable[[Optional[int]], None]] = None
        # run-to-run bit-identical training (the reference's Trainer runs with deterministic=True, pipeline_utils.py:373): the
        # multi-contributor reductions go through the fixed-point shadow of the gradient buffer (ops.set_deterministic).  On by
        # default (0.15-0.2 ms per step); IG_DETERMINISTIC=0 or ``engine.deterministic = False`` selects the float atomics.
        self.deterministic = os.environ.get("IG_DETERMINISTIC", "1") != "0"
        # training: the last stage's BatchNorm + ReLU run inside the classifier kernels (``fuse_tail = False``: the separate passes)
        self.fuse_tail = True
        self._det_pending: List[Tuple[int, int]] = []
        self._pos_cache: Dict[int, Any] = {}
        self._last: Optional[Dict[str, Any]] = None
        self._generation = 0  # bumped by every forward(save=True): the saved activations belong to exactly one forward

    # ---- helpers -------------------------------------------------------------------------------
    def P(self, name: str) -> torch.Tensor:  # fp32 master segment (biases, norm affine, classifier)
        return self.store.seg(self.store.flat, name)

    def Gd(self, name: str) -> torch.Tensor:
        return self.store.seg(self.store.grad, name)

    def W(self, name: str) -> BT:
        return self.store.w(name)

    def _need(self, last_name: Optional[str]) -> None:
        """The parameters up to and including ``last_name`` (flat = forward order; None: all) are about to be read."""
        if self.param_wait is not None:
            if last_name is None:
                self.param_wait(None)
            else:
                ent = self.store.entries[last_name]
                self.param_wait(ent.offset + ent.numel)

    def mark_params_changed(self) -> None:
        self.shadow_dirty = True
        self.shadow_t_dirty = True

    def _prepare_shadow(self) -> None:
        if self.shadow_dirty or self.store.shadow is None or self.store.shadow_split != self.split:
            if self.master_sync is not None:  # data parallel with sharded fp32 masters: complete them before they are read (collective)
                self.master_sync()
            self.store.refresh_shadow(self.split)
            self.shadow_dirty = False
            self.shadow_t_dirty = True

    def _prepare_shadow_t(self) -> None:
        st = self.store.shadow_t
        if self.shadow_t_dirty or st is None or st.split != self.split:
            self.store.refresh_shadow_t()
            self.shadow_t_dirty = False

    def geometry(self, size: int) -> SegConfig:
        """The configuration for a square input of ``size`` pixels: the configured one, or the same network on another token
        grid (the reference interpolates the position table and reshapes by the actual token count, pritvhi.py:149-203,
        model.py:406-413)."""
        if size == self.cfg.img_size:
            return self.cfg
        if size % self.cfg.patch or size <= 0:
            raise ValueError(f"image size {size} is not a multiple of the patch size {self.cfg.patch} (the reference would drop the border)")
        import dataclasses

        return dataclasses.replace(self.cfg, img_size=size)

    def pos_embed_for(self, cfg: SegConfig) -> torch.Tensor:
        """The (1, 1 + T g^2, D) position table of geometry ``cfg``: the checkpointed buffer, or -- for another input size -- its
        bicubic, align_corners=True interpolation exactly as ``interpolate_pos_encoding`` computes it (pritvhi.py:149-203),
        evaluated once per size in float32 on the host (the reference's CPU arithmetic) and cached until the parameters change."""
        base = self.buffers["prithvi_encoder.pos_embed"]
        if cfg.img_size == self.cfg.img_size:
            return base
        key = (cfg.img_size, base.data_ptr(), int(base._version))
        hit = self._pos_cache.get(cfg.img_size)
        if hit is not None and hit[0] == key:
            return hit[1]
        g0, T, D, pch = self.cfg.grid, cfg.num_frames, cfg.embed_dim, cfg.patch
        out = interpolate_pos_encoding(base.detach().float().cpu(), (T, g0, g0), (1, pch, pch), (T, cfg.img_size, cfg.img_size), D)
        out = out.contiguous().to(base.device)
        self._pos_cache[cfg.img_size] = (key, out)
        return out

    def workspace(self, B: int, training: bool, cfg: Optional[SegConfig] = None) -> Dict[str, Any]:
        cfg = cfg or self.cfg
        key = (B, training, self.split, str(self.store.flat.device), cfg.img_size)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        dev, sp = self.store.flat.device, self.split
        D, L, N, T, G = cfg.embed_dim, cfg.depth, cfg.tokens, cfg.num_frames, cfg.G
        M = B * N
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
        ws = {"B": B, "M": M}
        ws["patches"] = BT.empty((B * T * G, cfg.patch_k), sp, dev)
        nsave = L if training else 1
        ws["x_in"] = [f32(M, D) for _ in range(nsave + 1 if training else 2)]
        ws["x_mid"] = [f32(M, D) for _ in range(nsave)]
        ws["a"] = [BT.empty((M, D), sp, dev) for _ in range(nsave)]
        ws["qkv"] = [BT.empty((M, 3 * D), sp, dev) for _ in range(nsave)]
        ws["o"] = [BT.empty((M, D), sp, dev) for _ in range(nsave)]
        ws["c"] = [BT.empty((M, D), sp, dev) for _ in range(nsave)]
        ws["hact"] = [BT.empty((M, 4 * D), sp, dev) for _ in range(nsave)]
        # "hpre" holds gelu'(fc1 pre-activation), written by the fc1 epilogue and consumed by the fc2 dgrad epilogue
        ws["hpre"] = [BT.empty((M, 4 * D), sp, dev) for _ in range(nsave)] if training else None
        ws["lse"] = [f32(B, cfg.num_heads, N) for _ in range(nsave)] if training else None
        ws["mean1"] = [f32(M) for _ in range(nsave)]
        ws["rstd1"] = [f32(M) for _ in range(nsave)]
        ws["mean2"] = [f32(M) for _ in range(nsave)]
        ws["rstd2"] = [f32(M) for _ in range(nsave)]
        ws["meanF"], ws["rstdF"] = f32(M), f32(M)
        dims = cfg.head_dims
        hs = cfg.head_sizes  # (input side, ConvTranspose output side, Conv2d output side) per upscaling block
        ws["f"] = [BT.empty((B, cfg.grid, cfg.grid, dims[0]), sp, dev)] + [BT.empty((B, hs[i][2], hs[i][2], dims[i + 1]), sp, dev) for i in range(4)]
        ws["u"] = [BT.empty((B, hs[i][1], hs[i][1], dims[i + 1]), sp, dev) for i in range(4)]
        ws["cv"] = [BT.empty((B, hs[i][2], hs[i][2], dims[i + 1]), sp, dev) for i in range(4)]
        ws["bn_scale"] = [f32(dims[i + 1]) for i in range(4)]
        ws["bn_shift"] = [f32(dims[i + 1]) for i in range(4)]
        ws["bn_mean"] = [f32(dims[i + 1]) for i in range(4)]
        ws["bn_rstd"] = [f32(dims[i + 1]) for i in range(4)]
        ws["bn_sums"] = torch.empty(2 * max(dims), dtype=torch.float64, device=dev)
        if training:
            # backward scratch (reused across blocks / stages)
            ws["dx"] = f32(M, D)
            ws["dxb"] = BT.empty((M, D), sp, dev)
            ws["dxb2"] = BT.empty((M, D), sp, dev)
            ws["wgrad_groups"] = {}
            ws["det_folds"] = {}  # prepared range tables: deterministic folds per block / per backward, zeroing plan per optimizer range
            ws["wgrad8"] = {}  # block -> the grouped 8-phase weight-gradient kernel covers it
            ws["dtmp"] = BT.empty((M, D), sp, dev)
            ws["dh"] = BT.empty((M, 4 * D), sp, dev)
            ws["dqkv"] = BT.empty((M, 3 * D), sp, dev)
            ws["delta"] = f32(B * cfg.num_heads * N)
            ws["dpe"] = BT.empty((B * T * G, D), sp, dev)
            ws["df"] = [BT.empty(tuple(ws["f"][i].shape), sp, dev) for i in range(5)]
            ws["dcv"] = [BT.empty(tuple(ws["cv"][i].shape), sp, dev) for i in range(4)]
            ws["du"] = [BT.empty(tuple(ws["u"][i].shape), sp, dev) for i in range(4)]
        self._ws[key] = ws
        return ws

    # ---- forward -------------------------------------------------------------------------------
    def forward(self, img: torch.Tensor, training: bool, save: bool, out: Optional[torch.Tensor] = None,
                update_running: bool = True) -> torch.Tensor:
        """img (B,C,T,H,W) f32 [or (B,C,H,W) when T==1, pritvhi.py:507-509] -> logits (B,ncls,H,W) f32.

        ``training`` selects BatchNorm batch statistics + dropout (nn.Module.train()); ``save`` keeps the
        activations needed by :meth:`backward`.
        """
        cfg = self.cfg
        if img.dim() == 4 and cfg.num_frames == 1:
            img = img.unsqueeze(2)
        if img.dim() != 5 or img.shape[1] != cfg.in_chans or img.shape[2] != cfg.num_frames:
            raise ValueError(f"expected (B,{cfg.in_chans},{cfg.num_frames},H,W) input, got {tuple(img.shape)}")
        if img.shape[3] != img.shape[4]:
            raise ValueError(f"expected square chips, got {tuple(img.shape[3:])} (model.py:406-413 reshapes the tokens to a square grid)")
        cfg = self.geometry(int(img.shape[3]))
        if not img.is_cuda:
            raise ops._lib.HipLibraryError("PrithviSeg.forward needs a HIP device tensor: instageo_amd has no CPU path")
        img = img.contiguous().float()
        B = img.shape[0]
        if B == 0:  # empty batch: nothing to launch
            return torch.empty((0, cfg.num_classes, cfg.out_size, cfg.out_size), dtype=torch.float32, device=img.device)
        if training:  # BatchNorm batch statistics are a grid-wide reduction: the mode must be in place before the first forward
            self._sync_deterministic()
        ws = self.encoder_forward(img, save)
        logits = self._head_forward(ws, B, training, out, update_running, cfg)
        if save:
            self._generation += 1
        self._last = {"ws": ws, "B": B, "training": training, "generation": self._generation, "cfg": cfg} if save else None
        return logits

    def encoder_forward(self, img: torch.Tensor, save: bool = False) -> Dict[str, Any]:
        """Patch embed + L blocks + final LayerNorm (``PrithviViT.forward`` + the feature reshape, pritvhi.py:498-530,
        model.py:406-413) on a validated (B, C, T, H, W) f32 device batch; returns the workspace whose ``["f"][0]`` holds the
        (B, 14, 14, D*T) feature image.  ``bench.py`` times this leg alone (the north-star roofline is stated on it)."""
        cfg = self.geometry(int(img.shape[3]))
        B = img.shape[0]
        self._prepare_shadow()
        ws = self.workspace(B, save, cfg)
        D, L, N, T, G, H = cfg.embed_dim, cfg.depth, cfg.tokens, cfg.num_frames, cfg.G, cfg.num_heads
        M = B * N
        e = "prithvi_encoder."
        pos = self.pos_embed_for(cfg)
        ops.patchify(img, cfg.patch, ws["patches"])
        x = ws["x_in"][0]
        self._need(e + "patch_embed.proj.bias")
        ops.cls_rows(x, self.P(e + "cls_token"), pos, B, N, D)
        ops.patch_embed_fwd(ws["patches"], self.W(e + "patch_embed.proj.weight"), self.P(e + "patch_embed.proj.bias"), pos, x, B,
                            T * G, D, cfg.patch_k)
        for i in range(L):
            s = i if save else 0
            b = f"{e}blocks.{i}."
            x_in = ws["x_in"][i if save else i % 2]
            x_out = ws["x_in"][i + 1 if save else (i + 1) % 2]
            x_mid = ws["x_mid"][s]
            self._need(b + "mlp.fc2.bias")
            ops.layernorm_fwd(x_in, self.P(b + "norm1.weight"), self.P(b + "norm1.bias"), ws["a"][s], ws["mean1"][s], ws["rstd1"][s], M, D)
            ops.linear_fwd(ws["a"][s], self.W(b + "attn.qkv.weight"), self.P(b + "attn.qkv.bias"), ws["qkv"][s], M, 3 * D, D)
            ops.attention_fwd(ws["qkv"][s], ws["o"][s], ws["lse"][s] if save else None, B, N, H, cfg.head_dim)
            ops.linear_residual_fwd(ws["o"][s], self.W(b + "attn.proj.weight"), self.P(b + "attn.proj.bias"), x_in, x_mid, M, D, D)
            ops.layernorm_fwd(x_mid, self.P(b + "norm2.weight"), self.P(b + "norm2.bias"), ws["c"][s], ws["mean2"][s], ws["rstd2"][s], M, D)
            ops.linear_fwd(ws["c"][s], self.W(b + "mlp.fc1.weight"), self.P(b + "mlp.fc1.bias"), ws["hact"][s], M, 4 * D, D, act=1,
                           pre=ws["hpre"][s] if save else None)
            ops.linear_residual_fwd(ws["hact"][s], self.W(b + "mlp.fc2.weight"), self.P(b + "mlp.fc2.bias"), x_mid, x_out, M, D, 4 * D)
        x_fin = ws["x_in"][L if save else L % 2]
        self._need(None)  # final norm + decode head
        # final LayerNorm writes the (B, 14, 14, D*T) feature image directly (model.py:406-413, c = d*T + t)
        ops.layernorm_fwd(x_fin, self.P(e + "norm.weight"), self.P(e + "norm.bias"), ws["f"][0], ws["meanF"], ws["rstdF"], M, D,
                          feat_T=T, feat_G=G, ntok=N)
        return ws

    def _drop_counter(self, advance: bool) -> Optional[torch.Tensor]:
        dev = self.store.flat.device
        if self._drop_step is None or self._drop_step.device != dev:
            self._drop_step = torch.zeros(1, dtype=torch.int32, device=dev)
        if advance:
            self._drop_step += 5  # five dropout sites use seeds +0..+4
        return self._drop_step

    def _head_forward(self, ws, B: int, training: bool, out, update_running: bool, cfg: Optional[SegConfig] = None) -> torch.Tensor:
        cfg = cfg or self.cfg
        dims, hs, ks = cfg.head_dims, cfg.head_sizes, cfg.head_kernels
        p = cfg.drop_p if training else 0.0
        sd = self._drop_counter(advance=True) if p > 0 else None
        h = "segmentation_head."
        eval_tail_fused = False
        for i in range(4):
            Hs, Hu, Ho = hs[i]  # ConvTranspose2d: Hs -> Hu = 2 Hs; Conv2d(k, padding=1): Hu -> Ho = Hu + 3 - k (model.py:349-378)
            ops.convT_fwd(ws["f"][i], self.W(f"{h}{i}.0.weight"), self.P(f"{h}{i}.0.bias"), ws["u"][i], B, Hs, Hs, dims[i], dims[i + 1],
                          seed=self.drop_seed + i, p=p, seed_dev=sd)
            if not training:
                # eval mode: BatchNorm(running stats)+ReLU is a per-channel affine folded into the conv epilogue (one HBM pass less)
                ops.bn_eval_affine(self.P(f"{h}{i}.3.weight"), self.P(f"{h}{i}.3.bias"), self.buffers[f"{h}{i}.3.running_mean"],
                                   self.buffers[f"{h}{i}.3.running_var"], ws["bn_scale"][i], ws["bn_shift"][i], dims[i + 1])
                if i == 3 and ks[i] == 3 and self.fuse_tail:
                    # inference tail: where the direct 48-channel kernel runs, the classifier is applied in the last convolution's epilogue
                    # and the head's largest activation is neither written nor read again
                    if out is None:
                        out = torch.empty((B, cfg.num_classes, cfg.out_size, cfg.out_size), dtype=torch.float32, device=ws["f"][4].hi.device)
                    eval_tail_fused = ops.conv3x3_cls_fwd(ws["u"][i], self.W(f"{h}{i}.2.weight"), self.P(f"{h}{i}.2.bias"), ws["bn_scale"][i],
                                                          ws["bn_shift"][i], None, self.P(h + "5.weight"), self.P(h + "5.bias"), out, B, Hu, Hu,
                                                          dims[i + 1], cfg.num_classes)
                    if eval_tail_fused:
                        continue
                ops.conv_fwd(ws["u"][i], self.W(f"{h}{i}.2.weight"), self.P(f"{h}{i}.2.bias"), ws["f"][i + 1], B, Hu, Hu,
                             dims[i + 1], dims[i + 1], ks[i], bn_scale=ws["bn_scale"][i], bn_shift=ws["bn_shift"][i])
                continue
            if i == 3 and self.fuse_tail:
                # last stage: statistics only -- from the convolution's epilogue where the direct kernel runs -- and BatchNorm + ReLU are
                # applied inside the classifier kernels (the activation between them, the largest tensor of the head, is never written:
                # ig_classifier_bn_fwd / _bwd)
                bn_args = (self.P(f"{h}{i}.3.weight"), self.P(f"{h}{i}.3.bias"), self.buffers[f"{h}{i}.3.running_mean"],
                           self.buffers[f"{h}{i}.3.running_var"], ws["bn_scale"][i], ws["bn_shift"][i], ws["bn_mean"][i], ws["bn_rstd"][i])
                if ks[i] == 3 and ops.conv3x3_fwd_stats(ws["u"][i], self.W(f"{h}{i}.2.weight"), self.P(f"{h}{i}.2.bias"), ws["cv"][i], ws["bn_sums"],
                                                        B, Hu, Hu, dims[i + 1], dims[i + 1]):
                    ops.bn_finalize(ws["bn_sums"], *bn_args, B * Ho * Ho, dims[i + 1], training and update_running)
                else:
                    if ks[i] != 3:
                        ops.conv_fwd(ws["u"][i], self.W(f"{h}{i}.2.weight"), self.P(f"{h}{i}.2.bias"), ws["cv"][i], B, Hu, Hu, dims[i + 1],
                                     dims[i + 1], ks[i])
                    ops.bn_stats(ws["cv"][i], *bn_args, ws["bn_sums"], B * Ho * Ho, dims[i + 1], training and update_running)
                if training and update_running:
                    self.buffers[f"{h}{i}.3.num_batches_tracked"] += 1
                continue
            stats_ready = False  # the direct convolutions (48 / 96 channels) leave the BatchNorm statistics: no statistics pass
            if ks[i] == 3 and training:
                stats_ready = ops.conv3x3_fwd_stats(ws["u"][i], self.W(f"{h}{i}.2.weight"), self.P(f"{h}{i}.2.bias"), ws["cv"][i], ws["bn_sums"], B, Hu, Hu,
                                                    dims[i + 1], dims[i + 1])
            else:
                ops.conv_fwd(ws["u"][i], self.W(f"{h}{i}.2.weight"), self.P(f"{h}{i}.2.bias"), ws["cv"][i], B, Hu, Hu, dims[i + 1],
                             dims[i + 1], ks[i])
            ops.bn_relu_fwd(ws["cv"][i], self.P(f"{h}{i}.3.weight"), self.P(f"{h}{i}.3.bias"), self.buffers[f"{h}{i}.3.running_mean"],
                            self.buffers[f"{h}{i}.3.running_var"], ws["f"][i + 1], ws["bn_scale"][i], ws["bn_shift"][i], ws["bn_mean"][i],
                            ws["bn_rstd"][i], ws["bn_sums"], B * Ho * Ho, dims[i + 1], training, training and update_running, stats_ready=stats_ready)
            if training and update_running:
                self.buffers[f"{h}{i}.3.num_batches_tracked"] += 1
        S = cfg.out_size  # = img_size for every variant at its native chip size (3 x 3 kernels keep 2 x; 600M: 228 -> 224)
        if out is None:
            out = torch.empty((B, cfg.num_classes, S, S), dtype=torch.float32, device=ws["f"][4].hi.device)
        if eval_tail_fused:
            pass
        elif training and self.fuse_tail:
            ops.classifier_bn_fwd(ws["cv"][3], ws["bn_scale"][3], ws["bn_shift"][3], self.P(h + "5.weight"), self.P(h + "5.bias"), out, B, S * S,
                                  dims[4], cfg.num_classes, seed=self.drop_seed + 4, p=p, seed_dev=sd)
        else:
            ops.classifier_fwd(ws["f"][4], self.P(h + "5.weight"), self.P(h + "5.bias"), out, B, S * S, dims[4], cfg.num_classes,
                               seed=self.drop_seed + 4, p=p, seed_dev=sd)
        ws["tail_fused"] = bool(training and self.fuse_tail)
        return out

    def features_nchw(self) -> torch.Tensor:
        """reshaped_features of ``forward(..., return_features=True)``: (B, D*T, 14, 14) f32."""
        assert self._last is not None
        return self._last["ws"]["f"][0].float().permute(0, 3, 1, 2).contiguous()

    def _sync_deterministic(self) -> None:
        """Register / unregister this engine's gradient buffer with the library's deterministic-reduction mode (no-op when
        nothing changed; a change synchronises the device)."""
        if self.deterministic:
            ops.set_deterministic(self.store.ensure_grad())
        elif self.store.grad is not None and ops._DET["grad"] is self.store.grad:
            ops.set_deterministic(None)

    # ---- backward ------------------------------------------------------------------------------
    def backward(self, dlogits: torch.Tensor, count: Optional[torch.Tensor] = None, generation: Optional[int] = None,
                 fresh: bool = False) -> None:
        """Accumulate d loss / d params into the flat grad buffer from d loss / d logits.

        ``count``: optional device double[2] (``ig_ce_loss`` stats) whose [1] normalises un-normalised dlogits.
        ``generation``: the forward this backward belongs to (autograd bridge); the engine keeps ONE set of saved
        activations, so a backward after a newer grad-enabled forward would silently use the wrong ones -- it raises.
        ``fresh``: first backward of a step, after :meth:`zero_grads_for_step`: the Blocks' weight gradients are WRITTEN, not
        accumulated (their old contents are neither read nor had to be zeroed: 8 bytes per weight less HBM traffic per step).
        """
        assert self._last is not None, "forward(save=True) must precede backward"
        cfg = self._last["cfg"]
        if generation is not None and generation != self._last["generation"]:
            raise RuntimeError(
                "PrithviSeg: backward of a forward whose saved activations were overwritten by a later grad-enabled forward "
                "(one set of activations per engine: run backward before the next training forward, or use a second module)")
        ws, B, training = self._last["ws"], self._last["B"], self._last["training"]
        self.store.ensure_grad()
        self._sync_deterministic()
        self._det_pending = []
        D, L, N, T, G, H = cfg.embed_dim, cfg.depth, cfg.tokens, cfg.num_frames, cfg.G, cfg.num_heads
        M = B * N
        dims, hs, ks = cfg.head_dims, cfg.head_sizes, cfg.head_kernels
        p = cfg.drop_p if training else 0.0
        sd = self._drop_counter(advance=False) if p > 0 else None
        h = "segmentation_head."
        S = cfg.out_size
        if not training:
            raise RuntimeError("backward through eval-mode BatchNorm is not supported (reference trains in train mode)")
        fused = ws.get("tail_fused", False)
        if fused:
            ops.classifier_bn_bwd(dlogits.contiguous(), ws["cv"][3], ws["bn_scale"][3], ws["bn_shift"][3], ws["bn_mean"][3], ws["bn_rstd"][3],
                                  self.P(h + "5.weight"), ws["dcv"][3], self.Gd(h + "5.weight"), self.Gd(h + "5.bias"),
                                  self.Gd(f"{h}3.3.weight"), self.Gd(f"{h}3.3.bias"), ws["bn_sums"], count, B, S * S, dims[4], cfg.num_classes,
                                  seed=self.drop_seed + 4, p=p, seed_dev=sd)
        else:
            ops.classifier_bwd(dlogits.contiguous(), ws["f"][4], self.P(h + "5.weight"), ws["df"][4], self.Gd(h + "5.weight"),
                               self.Gd(h + "5.bias"), count, B, S * S, dims[4], cfg.num_classes, seed=self.drop_seed + 4, p=p, seed_dev=sd)
        for i in range(3, -1, -1):
            Hs, Hu, Ho = hs[i]
            Mo = B * Ho * Ho
            C1 = dims[i + 1]
            if not (fused and i == 3):
                ops.bn_relu_bwd(ws["cv"][i], ws["df"][i + 1], ws["bn_scale"][i], ws["bn_shift"][i], ws["bn_mean"][i], ws["bn_rstd"][i],
                                ws["dcv"][i], self.Gd(f"{h}{i}.3.weight"), self.Gd(f"{h}{i}.3.bias"), ws["bn_sums"], Mo, C1)
            # The Conv2d bias sits in front of a training-mode BatchNorm: its gradient, the pixel sum of the BatchNorm backward's
            # output, is ZERO by construction (sum(dy - mean(dy)) = 0 and sum(x_hat) = 0); the reference's autograd returns the
            # fp32 rounding noise of that sum (1e-8 here).  It is left at exactly 0: no column-sum pass over dcv (two launches per
            # step on the gather-GEMM stages).  The ConvTranspose bias gradients are real and ride on the weight-gradient kernels.
            ops.conv_wgrad(ws["dcv"][i], ws["u"][i], self.Gd(f"{h}{i}.2.weight"), B, Hu, Hu, C1, C1, ks[i], dbias=None)
            ops.conv_dgrad(ws["dcv"][i], self.W(f"{h}{i}.2.weight"), ws["du"][i], B, Hu, Hu, C1, C1, ks[i], seed=self.drop_seed + i, p=p, seed_dev=sd)
            ops.convT_wgrad(ws["du"][i], ws["f"][i], self.Gd(f"{h}{i}.0.weight"), B, Hs, Hs, dims[i], C1, dbias=self.Gd(f"{h}{i}.0.bias"))
            if i > 0 or not self.freeze_backbone:
                ops.convT_dgrad(ws["du"][i], self.W(f"{h}{i}.0.weight"), ws["df"][i], B, Hs, Hs, dims[i], C1)
        head0 = "segmentation_head.0.0.weight"
        self._grad_ready(head0, None, last=self.freeze_backbone)
        if self.freeze_backbone:
            return
        e = "prithvi_encoder."
        dx, dxb, dxb2 = ws["dx"], ws["dxb"], ws["dxb2"]
        self._prepare_shadow_t()
        WT = self.store.wt

        def block_start(i: int) -> str:
            return f"{e}blocks.{i}.norm1.weight" if i < L else e + "norm.weight"

        last_fc2_bias = self.Gd(f"{e}blocks.{L - 1}.mlp.fc2.bias") if L > 0 else None
        ops.layernorm_bwd(ws["df"][0], ws["x_in"][L], ws["meanF"], ws["rstdF"], self.P(e + "norm.weight"), dx, False, dxb,
                          self.Gd(e + "norm.weight"), self.Gd(e + "norm.bias"), last_fc2_bias, M, D, feat_T=T, feat_G=G, ntok=N)
        self._grad_ready(e + "norm.weight", head0)
        for i in range(L - 1, -1, -1):
            b = f"{e}blocks.{i}."
            # The four weight gradients of the block run as ONE grouped launch after its data gradients (ig_linear_wgrad_group):
            # their 108 output tiles (D = 768) fill the CUs with 2-3 token ranges per tile, where each GEMM alone needed 7-28
            # splits and as many partial tiles to fold.  dxb2 keeps the proj-side residual gradient alive next to the fc2-side one.
            # fc2: x_out = x_mid + hact @ W2^T + b2   (its bias grad came from the LayerNorm backward that produced dx;
            # the fc1 bias gradient = column sums of dh is fused into this dgrad's epilogue)
            ops.linear_dgrad(dxb, self.W(b + "mlp.fc2.weight"), ws["dh"], M, D, 4 * D, pre=ws["hpre"][i], colsum=self.Gd(b + "mlp.fc1.bias"),
                             wt=WT(i, "mlp.fc2.weight"))
            # fc1
            ops.linear_dgrad(ws["dh"], self.W(b + "mlp.fc1.weight"), ws["dtmp"], M, 4 * D, D, wt=WT(i, "mlp.fc1.weight"))
            ops.layernorm_bwd(ws["dtmp"], ws["x_mid"][i], ws["mean2"][i], ws["rstd2"][i], self.P(b + "norm2.weight"), dx, True, dxb2,
                              self.Gd(b + "norm2.weight"), self.Gd(b + "norm2.bias"), self.Gd(b + "attn.proj.bias"), M, D)
            # proj
            ops.linear_dgrad(dxb2, self.W(b + "attn.proj.weight"), ws["dtmp"], M, D, D, wt=WT(i, "attn.proj.weight"))
            # (the qkv bias gradient = column sums of dqkv rides on the attention backward: fused into the single-pass kernel)
            ops.attention_bwd(ws["qkv"][i], ws["o"][i], ws["dtmp"], ws["lse"][i], ws["delta"], ws["dqkv"], B, N, H, cfg.head_dim,
                              dbias=self.Gd(b + "attn.qkv.bias"))
            # qkv
            ops.linear_dgrad(ws["dqkv"], self.W(b + "attn.qkv.weight"), ws["dtmp"], M, 3 * D, D, wt=WT(i, "attn.qkv.weight"))
            grp = ws["wgrad_groups"].get(i)
            if grp is None or grp.items[0][2].data_ptr() != self.Gd(b + "mlp.fc2.weight").data_ptr():
                grp = ws["wgrad_groups"][i] = ops.WgradGroup([
                    (dxb, ws["hact"][i], self.Gd(b + "mlp.fc2.weight"), D, 4 * D),
                    (ws["dh"], ws["c"][i], self.Gd(b + "mlp.fc1.weight"), 4 * D, D),
                    (dxb2, ws["o"][i], self.Gd(b + "attn.proj.weight"), D, D),
                    (ws["dqkv"], ws["a"][i], self.Gd(b + "attn.qkv.weight"), 3 * D, D),
                ], M)  # prepared once per (workspace, block): pointers of the workspace and of the flat gradient buffer
            grp.launch(overwrite=fresh)
            # Did the grouped 8-phase kernel take it (ordered fold straight into dW), or the per-GEMM engines (whose split-K adds go
            # through the fixed-point shadow)?  The library decides on EVERY call (IG_WGRAD8, reserved CUs, a stream capture meeting an
            # unplanned shape ...), so the answer is read back every time: a change re-plans this block's fold and the zeroing table.
            took8 = ops.last_kernel().startswith(("gemm8w_kernel", "gemm4w_kernel"))
            if ws["wgrad8"].get(i) is not took8:
                ws["wgrad8"][i] = took8
                ws["det_folds"].pop(i, None)
                for k in [k for k in ws["det_folds"] if isinstance(k, tuple)]:
                    del ws["det_folds"][k]
                ws.pop("zero_small", None)
            if self.deterministic and i not in ws["det_folds"]:
                # the grouped kernel never writes through the shadow: the fold then only visits the small vectors between the four
                # weight matrices (the whole block otherwise)
                lo_b, hi_b = self.store.entries[block_start(i)].offset, self.store.entries[block_start(i + 1)].offset
                ws["det_folds"][i] = self._block_small_ranges(i, lo_b, hi_b) if ws["wgrad8"][i] else [(lo_b, hi_b)]
            prev_bias = self.Gd(f"{e}blocks.{i - 1}.mlp.fc2.bias") if i > 0 else None
            ops.layernorm_bwd(ws["dtmp"], ws["x_in"][i], ws["mean1"][i], ws["rstd1"][i], self.P(b + "norm1.weight"), dx, True, dxb,
                              self.Gd(b + "norm1.weight"), self.Gd(b + "norm1.bias"), prev_bias, M, D)
            self._grad_ready(block_start(i), block_start(i + 1), block=i)  # all grads of block i are final
        # patch embedding + cls token
        ops.patch_grad_prep(dx, ws["dpe"], self.Gd(e + "cls_token"), self.Gd(e + "patch_embed.proj.bias"), B, N, D)
        ops.linear_wgrad(ws["dpe"], ws["patches"], self.Gd(e + "patch_embed.proj.weight"), B * T * G, D, cfg.patch_k)
        self._grad_ready(e + "cls_token", block_start(0), last=True)
        self._plan_zero_small(ws)

    _BLOCK_WEIGHTS = ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight")

    def _block_small_ranges(self, i: int, lo: int, hi: int) -> List[Tuple[int, int]]:
        """Flat range [lo, hi) of Block ``i`` minus its four weight matrices."""
        b = f"prithvi_encoder.blocks.{i}."
        ranges, at = [], lo
        for ent in sorted((self.store.entries[b + n] for n in self._BLOCK_WEIGHTS), key=lambda e: e.offset):
            ranges.append((at, ent.offset))
            at = ent.offset + ent.numel
        ranges.append((at, hi))
        return [r for r in ranges if r[1] > r[0]]

    def zero_grads_for_step(self, lo: int, hi: int) -> None:
        """Clear flat gradient range [lo, hi) ahead of ``backward(..., fresh=True)``.  Once a backward pass has shown that the
        grouped weight-gradient kernel covers every Block, the Blocks' weight matrices are left out (the fresh backward overwrites
        them): one table-driven launch over the small vectors of the Blocks and a plain fill of what lies outside them."""
        g = self.store.ensure_grad()
        ws = self._last["ws"] if self._last is not None else None
        plan = ws.get("zero_small") if ws is not None and not self.freeze_backbone else None
        if plan is None or lo > plan[1] or hi < plan[2]:
            g[lo:hi].zero_()
            return
        plan[0].launch(g)
        if plan[1] > lo:
            g[lo : plan[1]].zero_()
        if hi > plan[2]:
            g[plan[2] : hi].zero_()

    def _plan_zero_small(self, ws) -> None:
        """After a backward pass: (table of the Blocks' small gradient ranges, first Block offset, end of the Blocks) when every
        Block's weight gradients went through the grouped kernel -- built here, outside any later graph capture."""
        L = self.cfg.depth
        if "zero_small" in ws or L == 0 or not all(ws["wgrad8"].get(i) for i in range(L)):
            return
        e = "prithvi_encoder."
        starts = [self.store.entries[f"{e}blocks.{i}.norm1.weight"].offset for i in range(L)] + [self.store.entries[e + "norm.weight"].offset]
        small: List[Tuple[int, int]] = []
        for i in range(L):
            small += self._block_small_ranges(i, starts[i], starts[i + 1])
        ws["zero_small"] = (ops.ZeroRanges(small, self.store.flat.device), starts[0], starts[L])

    def _grad_ready(self, first: str, until: Optional[str], block: Optional[int] = None, last: bool = False) -> None:
        """Tell the data-parallel layer that grads of flat range [offset(first), offset(until)) are final."""
        lo = self.store.entries[first].offset
        hi = self.store.total if until is None else self.store.entries[until].offset
        if self.deterministic:  # the shadow sums of the range become part of the gradients before anyone reads them
            ws = self._last["ws"]
            ranges = (ws["det_folds"].get(block) if block is not None else None) or [(lo, hi)]
            small = [r for r in ranges if r[1] - r[0] <= 65536]
            for r in ranges:
                if r[1] - r[0] > 65536:
                    ops.det_fold(*r)
            if self.on_grad_ready is None and not last:
                self._det_pending.extend(small)  # one process: a single table launch at the end of the backward pass
            else:
                key = ("table", tuple(self._det_pending + small))
                self._det_pending = []
                table = ws["det_folds"].get(key)
                if table is None:
                    table = ws["det_folds"][key] = ops.DetFoldRanges(key[1], self.store.flat.device)
                table.launch()
        if self.on_grad_ready is not None and hi > lo:
            self.on_grad_ready(lo, hi)


# --------------------------------------------------------------------------------------------------
# autograd bridge + nn.Module with the reference's contract
# --------------------------------------------------------------------------------------------------
# ``torch.ops.instageo_mi355x.prithvi_seg`` (torch_ops.py) finds the network behind an integer handle: a custom op takes tensors and
# scalars only.  Weak references: a handle dies with its module.
_NETWORKS: "weakref.WeakValueDictionary[int, PrithviSeg]" = weakref.WeakValueDictionary()
_HANDLES = itertools.count(1)


def network_of(handle: int) -> "PrithviSeg":
    net = _NETWORKS.get(int(handle))
    if net is None:
        raise RuntimeError(f"instageo_mi355x::prithvi_seg: no live PrithviSeg behind handle {handle}")
    return net


class _Holder(nn.Module):
    """Name-only container so that parameter paths equal the reference's state_dict keys."""


class PrithviSeg(nn.Module):
    """Prithvi Segmentation Model (drop-in for ``instageo.model.model.PrithviSeg``, model.py:292-419).

    Same constructor arguments, ``forward(img, return_features=False)`` semantics, attribute names
    (``prithvi_encoder``, ``segmentation_head``, ``model_args``) and ``state_dict()`` keys/shapes.
    Extra keyword ``precision`` ("bf16" | "bf16x3") and ``device``.
    """

    def __init__(
        self,
        temporal_step: int = 1,
        image_size: int = 224,
        num_classes: int = 2,
        load_pretrained_weights: bool = True,
        freeze_backbone: bool = True,
        model_bands: List[int] = list(range(6)),
        variant: str = "prithvi_eo_v1_100",
        embed_dims: Optional[List[int]] = None,
        depth: int = -1,
        precision: str = "bf16",
        device: Optional[Any] = None,
        **kwargs: Any,
    ) -> None:
        super().__init__()
        if kwargs:
            raise TypeError(f"unsupported PrithviSeg arguments: {sorted(kwargs)}")
        in_chans = 6 * max(1, len(model_bands) // 6)  # model.py:330 PRETRAINED_BANDS * (len(model_bands)//6)
        cfg = make_seg_config(variant, temporal_step, image_size, num_classes, depth, in_chans, embed_dims)
        self.cfg = cfg
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        device = torch.device(device)
        self.store = ParamStore(cfg, device)
        self._buffers_flat: Dict[str, torch.Tensor] = {}
        self.model_args = {
            "img_size": image_size, "num_frames": temporal_step, "patch_size": [1, cfg.patch, cfg.patch], "in_chans": in_chans,
            "embed_dim": cfg.embed_dim, "depth": cfg.depth, "num_heads": cfg.num_heads, "mlp_ratio": cfg.mlp_ratio,
        }  # fmt: skip
        self._build_tree()
        self.engine = SegEngine(cfg, self.store, self._buffers_flat, precision)
        self._handle = next(_HANDLES)
        _NETWORKS[self._handle] = self
        self.reset_parameters()
        if load_pretrained_weights:
            raise RuntimeError(
                "load_pretrained_weights=True needs the Hugging Face hub (model.py:221-251); this build has no network. "
                "Construct with load_pretrained_weights=False and call load_state_dict() / load_prithvi_checkpoint()."
            )
        self.freeze_backbone = bool(freeze_backbone)
        if freeze_backbone:  # model.py:341-343
            for p in self.prithvi_encoder.parameters():
                p.requires_grad = False
        self.engine.freeze_backbone = self.freeze_backbone

    # ---- module tree ---------------------------------------------------------------------------
    def _build_tree(self) -> None:
        cfg = self.cfg
        dev = self.store.flat.device

        def attach(path: str, tensor: torch.Tensor, is_param: bool) -> None:
            parts = path.split(".")
            mod: nn.Module = self
            for part in parts[:-1]:
                if part not in mod._modules:
                    mod.add_module(part, _Holder())
                mod = mod._modules[part]
            if is_param:
                mod.register_parameter(parts[-1], nn.Parameter(tensor))
            else:
                mod.register_buffer(parts[-1], tensor)

        # order matters: state_dict key order follows registration order (reference order)
        e = "prithvi_encoder."
        attach(e + "cls_token", self.store.entries[e + "cls_token"].api_view(self.store.flat), True)
        pos = torch.zeros(1, cfg.tokens, cfg.embed_dim, dtype=torch.float32, device=dev)
        attach(e + "pos_embed", pos, False)
        self._buffers_flat[e + "pos_embed"] = pos
        for name, ent in self.store.entries.items():
            if name == e + "cls_token":
                continue
            attach(name, ent.api_view(self.store.flat), True)
            if name == e + "patch_embed.proj.bias" and cfg.variant in TL_VARIANTS:
                # the reference's (unused) coordinate encoders: one trainable scale each, registered right after the patch
                # embedding (pritvhi.py:431-437).  They live OUTSIDE the flat buffer: no kernel reads them, they get no gradient,
                # and like any gradient-less parameter under torch.optim.AdamW they are never stepped (no weight decay either).
                attach(e + "temporal_embed_enc.scale", torch.full((1,), 0.1, dtype=torch.float32, device=dev), True)
                attach(e + "location_embed_enc.scale", torch.full((1,), 0.1, dtype=torch.float32, device=dev), True)
            if name.startswith("segmentation_head.") and name.endswith(".3.bias"):
                base = name[: -len("bias")]
                c = ent.shape[0]
                for bn, val in (("running_mean", torch.zeros(c, device=dev)), ("running_var", torch.ones(c, device=dev)),
                                ("num_batches_tracked", torch.zeros((), dtype=torch.int64, device=dev))):  # fmt: skip
                    attach(base + bn, val, False)
                    self._buffers_flat[base + bn] = val

    def _flat_params(self):
        if getattr(self, "_param_list", None) is None:
            named = dict(self.named_parameters())
            self._param_list = [(name, named[name]) for name in self.store.entries]
        return self._param_list

    def _apply(self, fn, recurse=True):
        """Keep parameters as views of the flat buffer across .to()/.cuda()."""
        new_flat = fn(self.store.flat)
        if new_flat.dtype != torch.float32:
            raise TypeError("PrithviSeg master parameters must stay float32 (bf16 operands are managed internally)")
        moved = new_flat.data_ptr() != self.store.flat.data_ptr() or new_flat.device != self.store.flat.device
        if moved:
            self.store.flat = new_flat.contiguous()
            self.store.grad = None
            self.store.shadow = None
            named = dict(self.named_parameters())
            for name, ent in self.store.entries.items():
                p = named[name]
                p.data = ent.api_view(self.store.flat)
                p.grad = None
            for name, p in named.items():  # parameters outside the flat buffer (the _tl variants' unused scales)
                if name not in self.store.entries:
                    p.data = fn(p.data)
            for name in list(self._buffers_flat):
                parts = name.split(".")
                mod = self
                for part in parts[:-1]:
                    mod = mod._modules[part]
                nb = fn(mod._buffers[parts[-1]])
                mod._buffers[parts[-1]] = nb
                self._buffers_flat[name] = nb
            self.engine._ws.clear()
            self.engine.mark_params_changed()
        return self

    # ---- init / checkpoint ---------------------------------------------------------------------
    @torch.no_grad()
    def reset_parameters(self) -> None:
        """Reference initialisation (pritvhi.py:463-477; head layers keep PyTorch defaults, model.py:360-378)."""
        cfg = self.cfg
        e = "prithvi_encoder."
        sd = dict(self.named_parameters())
        pe = get_3d_sincos_pos_embed(cfg.embed_dim, (cfg.num_frames, cfg.grid, cfg.grid), cls_token=True)
        self._buffers_flat[e + "pos_embed"].copy_(torch.from_numpy(pe).float().unsqueeze(0))
        for name, p in sd.items():
            if name.startswith(e):
                if name.endswith("cls_token"):
                    nn.init.normal_(p, std=0.02)
                elif name.endswith("patch_embed.proj.weight"):
                    w = torch.empty(p.shape[0], int(np.prod(p.shape[1:])))
                    nn.init.xavier_uniform_(w)
                    p.copy_(w.view(p.shape))
                elif name.endswith("patch_embed.proj.bias"):  # nn.Conv3d default bias init (untouched by init_weights)
                    bound = 1 / math.sqrt(cfg.patch_k)
                    p.copy_(torch.empty(p.shape).uniform_(-bound, bound))
                elif p.dim() == 2:
                    w = torch.empty(p.shape)
                    nn.init.xavier_uniform_(w)
                    p.copy_(w)
                elif ".norm" in name and name.endswith("weight"):
                    p.fill_(1.0)
                elif name.endswith("_embed_enc.scale"):  # TemporalEncoder / LocationEncoder trainable scale (pritvhi.py:289-290)
                    p.fill_(0.1)
                else:
                    p.zero_()
            else:
                if p.dim() == 4:  # kaiming_uniform_(a=sqrt(5)) as nn.Conv2d / nn.ConvTranspose2d.reset_parameters
                    w = torch.empty(p.shape)
                    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                    p.copy_(w)
                elif name.endswith(".3.weight"):
                    p.fill_(1.0)
                elif name.endswith(".3.bias"):
                    p.zero_()
                else:  # conv biases: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                    wname = name[: -len("bias")] + "weight"
                    wshape = sd[wname].shape
                    fan_in = wshape[1] * wshape[2] * wshape[3]
                    bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
                    p.copy_(torch.empty(p.shape).uniform_(-bound, bound))
        self.engine.mark_params_changed() if hasattr(self, "engine") else None

    def state_dict(self, *args, **kwargs):
        """``nn.Module.state_dict`` with the reference's keys.  Under sharded data parallelism (``zero1`` publishing the bf16 operand
        copy) a rank holds CURRENT fp32 masters only for the slices it owns between two ``sync_master_params()`` calls: reading them
        then would save mixed-age weights, and completing them is a collective that one rank cannot run alone -- so this raises."""
        mc = getattr(self.engine, "master_complete", None) if hasattr(self, "engine") else None
        if mc is not None and not mc():
            raise RuntimeError("PrithviSeg.state_dict(): the fp32 master parameters are sharded over the data-parallel ranks and this "
                               "rank's copy is incomplete; call PrithviSegmentationModule.sync_master_params() on EVERY rank first")
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        self.engine.mark_params_changed()
        return out

    def params_changed(self) -> None:
        """Call after modifying parameters outside this package (e.g. a foreign optimizer step)."""
        self.engine.mark_params_changed()

    # ---- forward -------------------------------------------------------------------------------
    def forward(self, img: torch.Tensor, return_features: bool = False):
        """(B,C,T,H,W) [or (B,C,H,W) if T==1] -> logits (B,num_classes,H,W) [, features (B,D*T,14,14)].

        The whole network is ONE dispatcher op, ``torch.ops.instageo_mi355x.prithvi_seg`` (torch_ops.py: CUDA/HIP implementation = the
        engine's forward, ``register_autograd`` backward = the engine's backward through ``prithvi_seg_backward``, a fake
        implementation for tracing), so the module sits behind a PyTorch custom op (north_star; base.py:28,69-77, model.py:392-419) and
        ``torch.compile(net, fullgraph=True)`` captures one node.  The parameters are passed as a tensor list -- autograd sees the
        dependence and ``loss.backward()`` fills ``p.grad`` -- and the activations stay in the engine's workspaces."""
        if not img.is_cuda:
            raise ops._lib.HipLibraryError("PrithviSeg.forward needs a HIP device tensor: instageo_amd has no CPU path")
        from . import torch_ops

        torch_ops.register()
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        params = [p for _, p in self._flat_params()]
        logits, feats, _gen = torch.ops.instageo_mi355x.prithvi_seg(img, params, self._handle, self.training, needs_grad, return_features)
        if return_features:
            return logits, feats
        return logits


def load_prithvi_checkpoint(model: PrithviSeg, state_dict: Dict[str, torch.Tensor], pretrained_bands: Optional[List[int]] = None,
                            model_bands: Optional[List[int]] = None) -> None:
    """Load a Prithvi MAE/ViT checkpoint into ``model.prithvi_encoder`` the way ``create_prithvi`` does (model.py:221-251):
    ``checkpoint_filter_fn_vit`` (instageo/model/utils.py:271-315, mirrored in :mod:`instageo_amd.utils`) drops decoder /
    mask-token keys, strips the ``encoder.`` prefix, keeps the model's own fixed ``pos_embed`` and selects the patch-embedding
    bands; blocks beyond ``depth`` are dropped (the reference builds the truncated model and loads non-strictly)."""
    from .utils import checkpoint_filter_fn_vit, encoder_view, get_state_dict

    view = encoder_view(model)
    enc = view.state_dict()
    nb = model.cfg.in_chans
    pretrained_bands = list(range(6)) if pretrained_bands is None else list(pretrained_bands)
    model_bands = (pretrained_bands * max(1, nb // len(pretrained_bands)))[:nb] if model_bands is None else list(model_bands)
    clean = checkpoint_filter_fn_vit(dict(get_state_dict(state_dict)), view, pretrained_bands, model_bands)
    clean = {k: v for k, v in clean.items() if not (k.startswith("blocks.") and int(k.split(".")[1]) >= model.cfg.depth)}
    missing = set(enc) - set(clean)
    unexpected = set(clean) - set(enc)
    if missing or unexpected:
        raise RuntimeError(f"checkpoint mismatch: missing {sorted(missing)[:5]}, unexpected {sorted(unexpected)[:5]}")
    full = model.state_dict()
    for k, v in clean.items():
        full["prithvi_encoder." + k] = v
    model.load_state_dict(full, strict=True)
