"""Thin torch-tensor wrappers over the C-ABI (``include/instageo_hip.h``).

PyTorch is plumbing only: it owns device memory and the current HIP stream; every arithmetic op on
the hot path is a HIP kernel reached through :func:`instageo_amd._lib.call`.  All wrappers require
contiguous CUDA(HIP) tensors and raise otherwise -- there is no CPU fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

BF16 = torch.bfloat16


class BT:
    """A bf16 device tensor, optionally *split* (hi + lo) for the bf16x3 precision mode."""

    __slots__ = ("hi", "lo")

    def __init__(self, hi: torch.Tensor, lo: Optional[torch.Tensor] = None):
        assert hi.dtype == BF16 and (lo is None or (lo.dtype == BF16 and lo.shape == hi.shape))
        self.hi, self.lo = hi, lo

    @staticmethod
    def _alloc(shape, split: bool, device, zero: bool) -> "BT":
        """hi [, lo] of one allocation, lo a fixed distance (a multiple of 256 bytes) ABOVE hi: the paired split-mode kernels (gemm8.hip
        NSEG = 2) fetch a K-tile's hi and lo chunks with ONE LDS-DMA instruction and carry that distance in the lo lanes' 32-bit offsets.
        Slices of both halves at the same position keep the distance."""
        make = torch.zeros if zero else torch.empty
        if not split:
            return BT(make(shape, dtype=BF16, device=device))
        shape = tuple(shape) if isinstance(shape, (tuple, list, torch.Size)) else (int(shape),)
        n = 1
        for d in shape:
            n *= int(d)
        pitch = (n + 127) // 128 * 128
        buf = make((2 * pitch,), dtype=BF16, device=device)
        return BT(buf[:n].view(shape), buf[pitch : pitch + n].view(shape))

    @staticmethod
    def empty(shape, split: bool, device) -> "BT":
        return BT._alloc(shape, split, device, False)

    @staticmethod
    def zeros(shape, split: bool, device) -> "BT":
        return BT._alloc(shape, split, device, True)

    @staticmethod
    def from_float(x: torch.Tensor, split: bool) -> "BT":
        x = x.contiguous().float()
        out = BT.empty(x.shape, split, x.device)
        split_bf16(x, out)
        return out

    @property
    def split(self) -> bool:
        return self.lo is not None

    @property
    def shape(self):
        return self.hi.shape

    def view(self, *shape) -> "BT":
        return BT(self.hi.view(*shape), None if self.lo is None else self.lo.view(*shape))

    def float(self) -> torch.Tensor:
        out = torch.empty(self.hi.shape, dtype=torch.float32, device=self.hi.device)
        _lib.call("ig_merge_bf16", _p(self.hi), _p(self.lo), _p(out), out.numel(), _stream())
        return out


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.HipLibraryError("instageo_amd ops need HIP device tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise _lib.HipLibraryError("instageo_amd ops need contiguous tensors")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---- optional per-launch HIP-event profiling (bench.py: roofline of the dominant kernels) ----------------
# PROF maps entry point -> {"work": flops or bytes summed over bracketed launches, "events": [(start, end, kernel), ...]}.
# torch.cuda.Event records on torch's current stream, which is the stream every kernel here is launched on.  Every profiled
# launch is also counted under the name of the KERNEL it started (``ig_last_kernel``: rocprofv3's name minus namespaces).
PROF = None
PROF_STRIDE = 1
PROF_KERNELS = None


def profile_begin(names, stride: int = 1) -> None:
    """Start per-launch event profiling of the named entry points.

    ``stride`` > 1 brackets only every stride-th launch of each entry point: an event pair drains the queue around the
    kernel (a few microseconds of bubble per pair), so inside a timed region only a sample is bracketed.  Pick a stride
    coprime with the per-layer launch pattern (7 is) so the sample keeps the mix of shapes.
    """
    global PROF, PROF_STRIDE, PROF_KERNELS
    PROF = {n: {"work": 0.0, "events": [], "seen": 0} for n in names}
    PROF_KERNELS = {}
    PROF_STRIDE = max(1, int(stride))


def profile_end():
    """Synchronise and return ``{"ops": {entry point: (bracketed launches, total_ms, total_work)}, "kernels": {kernel name:
    {"op", "calls" (all launches seen), "n" (bracketed), "ms", "work"}}}``."""
    global PROF, PROF_KERNELS
    prof, kern, PROF, PROF_KERNELS = PROF, PROF_KERNELS, None, None
    torch.cuda.synchronize()
    ops_out = {}
    for n, d in prof.items():
        tot = 0.0
        for a, b, k, w in d["events"]:
            ms = a.elapsed_time(b)
            tot += ms
            r = kern[k]
            r["n"] += 1
            r["ms"] += ms
            r["work"] += w
        ops_out[n] = (len(d["events"]), tot, d["work"])
    return {"ops": ops_out, "kernels": kern}


def _kernel_label(name: str) -> str:
    k = (_lib.load().ig_last_kernel() or b"").decode()
    return k or name


def _call(name: str, work: float, *args, entry: Optional[str] = None) -> None:
    """Call entry point ``entry`` (default ``name``); ``name`` is the key the launch is profiled under."""
    entry = entry or name
    if PROF is None or name not in PROF:
        _lib.call(entry, *args)
        return
    d = PROF[name]
    d["seen"] += 1
    bracket = (d["seen"] - 1) % PROF_STRIDE == 0
    if bracket:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
    _lib.load().ig_note_reset()
    _lib.call(entry, *args)
    k = _kernel_label(name)
    r = PROF_KERNELS.get(k)
    if r is None:
        r = PROF_KERNELS[k] = {"op": name, "calls": 0, "n": 0, "ms": 0.0, "work": 0.0}
    r["calls"] += 1
    if bracket:
        b.record()
        d["events"].append((a, b, k, work))
        d["work"] += work


def reserved_cus() -> int:
    return int(_lib.load().ig_get_reserved_cus())


def set_reserved_cus(n: int) -> None:
    """Compute units the persistent GEMM kernels leave free (for RCCL's kernels under data parallelism)."""
    _lib.call("ig_set_reserved_cus", int(n))


# ---- run-to-run deterministic reductions (include/instageo_hip.h: ig_set_deterministic) --------------------------------
_DET = {"grad": None, "shadow": None}


def set_deterministic(grad_flat: Optional[torch.Tensor]) -> None:
    """Register ``grad_flat`` (the flat fp32 gradient buffer) for order-independent reductions, or switch the mode off (None).

    While registered, the kernels add their bias / norm / head gradient contributions as 2^44 fixed-point integers into an int64
    shadow of the buffer; ``det_fold(lo, hi)`` adds the shadow into the gradients.  One buffer per process at a time.
    """
    if grad_flat is None:
        if _DET["grad"] is not None:
            torch.cuda.synchronize()
            _lib.call("ig_set_deterministic", None, None, 0, _stream())
        _DET["grad"] = _DET["shadow"] = None
        return
    g = _f32(grad_flat)
    if _DET["grad"] is not None and _DET["grad"].data_ptr() == g.data_ptr() and _DET["grad"].numel() == g.numel():
        return
    torch.cuda.synchronize()
    shadow = torch.zeros(g.numel(), dtype=torch.int64, device=g.device)
    _lib.call("ig_set_deterministic", _p(shadow), _p(g), g.numel(), _stream())
    _DET["grad"], _DET["shadow"] = g, shadow


def last_kernel() -> str:
    """Name of the kernel the most recent MFMA entry point of this thread launched (``ig_last_kernel``)."""
    return (_lib.load().ig_last_kernel() or b"").decode()


def deterministic() -> bool:
    return bool(_lib.load().ig_get_deterministic())


def det_fold(lo: int, hi: int) -> None:
    """Add the fixed-point shadow sums of flat gradient range [lo, hi) into the gradients and clear them (no-op when off)."""
    if _DET["grad"] is not None and hi > lo:
        _lib.call("ig_det_fold", int(lo), int(hi), _stream())


class ZeroRanges:
    """Prepared argument block of one ``ig_zero_ranges`` launch: ``base[lo:hi] = 0`` for a device table of flat ranges."""

    __slots__ = ("n", "ranges", "_table", "_longest")

    def __init__(self, ranges, device):
        self.ranges = [(int(a), int(b)) for a, b in ranges if b > a]
        self.n = len(self.ranges)
        self._table = torch.tensor(self.ranges or [(0, 0)], dtype=torch.int64).to(device)
        self._longest = max([b - a for a, b in self.ranges], default=0)

    def launch(self, base: torch.Tensor) -> None:
        if self.n:
            _lib.call("ig_zero_ranges", _p(_f32(base)), self.n, _p(self._table), self._longest, _stream())


class DetFoldRanges:
    """Prepared argument block of one ``ig_det_fold_ranges`` launch: a device table of flat ranges [(lo, hi), ...]."""

    __slots__ = ("n", "ranges", "_table", "_longest")

    def __init__(self, ranges, device):
        self.ranges = [(int(a), int(b)) for a, b in ranges if b > a]
        self.n = len(self.ranges)
        self._table = torch.tensor(self.ranges or [(0, 0)], dtype=torch.int64).to(device)
        self._longest = max([b - a for a, b in self.ranges], default=0)

    def launch(self) -> None:
        if _DET["grad"] is not None and self.n:
            _lib.call("ig_det_fold_ranges", self.n, _p(self._table), self._longest, _stream())


def _f32(t: torch.Tensor) -> torch.Tensor:
    assert t.dtype == torch.float32, t.dtype
    return t


# ------------------------------------------------------------------------------------------------------
def split_bf16(src: torch.Tensor, out: BT) -> None:
    _lib.call("ig_split_bf16", _p(_f32(src)), _p(out.hi), _p(out.lo), src.numel(), _stream())


def normalize_chips(src: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, temporal: int,
                    constant_multiplier: Optional[float] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B, T*C, H, W) int16|f32 -> (B, C, T, H, W) f32 normalised  (dataloader.py:495-524)."""
    B, TC, H, W = src.shape
    C = TC // temporal
    assert C * temporal == TC and mean.numel() == C and std.numel() == C
    dt = {torch.int16: 0, torch.float32: 1}[src.dtype]
    if out is None:
        out = torch.empty((B, C, temporal, H, W), dtype=torch.float32, device=src.device)
    mult = 1.0 if constant_multiplier is None else float(constant_multiplier)
    _call("ig_normalize_chips", float(src.numel()) * (src.element_size() + 4), _p(src), dt, _p(_f32(mean)), _p(_f32(std)), mult,
          int(constant_multiplier is not None), _p(out), B, temporal, C, H, W, _stream())
    return out


def crop_flip_normalize(src: torch.Tensor, params: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, temporal: int, im: int,
                        constant_multiplier: Optional[float] = None, labels: Optional[torch.Tensor] = None,
                        out: Optional[torch.Tensor] = None):
    """(B, T*C, Hs, Ws) int16|f32 -> (B, C, T, im, im) f32: per-chip crop at params[b] = (top, left) + optional
    hflip/vflip (params[b, 2:4]) + normalisation in one kernel; ``labels`` (B, Hs, Ws) f32 follow (dataloader.py:58-141, 527-585)."""
    B, TC, Hs, Ws = src.shape
    C = TC // temporal
    assert C * temporal == TC and mean.numel() == C and std.numel() == C
    assert params.dtype == torch.int32 and params.shape == (B, 4)
    dt = {torch.int16: 0, torch.float32: 1}[src.dtype]
    if out is None:
        out = torch.empty((B, C, temporal, im, im), dtype=torch.float32, device=src.device)
    lab_out = None
    if labels is not None:
        labels = _f32(labels)
        assert labels.shape == (B, Hs, Ws)
        lab_out = torch.empty((B, im, im), dtype=torch.float32, device=src.device)
    mult = 1.0 if constant_multiplier is None else float(constant_multiplier)
    work = float(B) * TC * im * im * (src.element_size() + 4) + (float(B) * im * im * 8 if labels is not None else 0.0)
    _call("ig_crop_flip_normalize", work, _p(src), dt, _p(_f32(mean)), _p(_f32(std)), mult, int(constant_multiplier is not None), _p(params),
          _p(out), _p(labels), _p(lab_out), B, temporal, C, Hs, Ws, im, _stream())
    return out, lab_out


def aug_rotate(buf: torch.Tensor, params: torch.Tensor, fill: float, labels: Optional[torch.Tensor] = None, label_fill: float = -1.0):
    """RandomRotation of a raw-domain batch (B, T*C, S, S) f32 [+ labels (B, S, S) f32]: nearest neighbour about the centre with
    constant fill, params (B, 8) int32 = {apply, 16.16 fixed-point inverse affine} (dataloader.py:144-187).  Returns new tensors."""
    B, CT, S, S2 = buf.shape
    assert S == S2 and buf.dtype == torch.float32 and params.dtype == torch.int32 and params.shape == (B, 8)
    out = torch.empty_like(buf)
    lab_out = None
    if labels is not None:
        labels = _f32(labels)
        assert labels.shape == (B, S, S)
        lab_out = torch.empty_like(labels)
    _call("ig_aug_rotate", float(buf.numel() + (0 if labels is None else labels.numel())) * 8, _p(buf), _p(out), _p(labels), _p(lab_out),
          _p(params), float(fill), float(label_fill), B, CT, S, _stream())
    return out, lab_out


def aug_brightness_contrast(buf: torch.Tensor, params: torch.Tensor, max_pixel: float) -> torch.Tensor:
    """RandomBrightnessContrast in place: params (B, 4) f32 = {apply, bright, contrast, 0} (dataloader.py:190-260)."""
    B, CT, S, _ = buf.shape
    assert buf.dtype == torch.float32 and params.dtype == torch.float32 and params.shape == (B, 4)
    _call("ig_aug_brightness_contrast", float(buf.numel()) * 12, _p(buf), _p(params), float(max_pixel), B, CT, S, _stream())
    return buf


def aug_blur(buf: torch.Tensor, apply: torch.Tensor, kernel2d: torch.Tensor, max_pixel: float) -> torch.Tensor:
    """RandomGaussianBlur: apply (B,) int32, kernel2d (k, k) f32 (dataloader.py:263-333).  Returns a new tensor."""
    B, CT, S, _ = buf.shape
    assert buf.dtype == torch.float32 and apply.dtype == torch.int32 and apply.numel() == B
    k = kernel2d.shape[0]
    out = torch.empty_like(buf)
    _call("ig_aug_blur", float(buf.numel()) * 8, _p(buf), _p(out), _p(apply), _p(_f32(kernel2d)), k, float(max_pixel), B, CT, S, _stream())
    return out


def aug_noise(buf: torch.Tensor, params: torch.Tensor, noise_std: float, max_pixel: float, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """RandomGaussianNoise in place: params (B, 2) int32 = {apply, seed}; ``noise`` = optional standard-normal field of buf's
    shape, else generated on the device from the per-chip seed (dataloader.py:336-386)."""
    B, CT, S, _ = buf.shape
    assert buf.dtype == torch.float32 and params.dtype == torch.int32 and params.shape == (B, 2)
    if noise is not None:
        noise = _f32(noise)
        assert noise.shape == buf.shape
    _call("ig_aug_noise", float(buf.numel()) * (8 if noise is None else 12), _p(buf), _p(params), _p(noise), float(noise_std), float(max_pixel),
          B, CT, S, _stream())
    return buf


def normalize_windows(tile: torch.Tensor, origins: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, temporal: int, crop: int,
                      constant_multiplier: Optional[float] = None, labels: Optional[torch.Tensor] = None,
                      out: Optional[torch.Tensor] = None):
    """Sliding-window gather + normalise in ONE launch: tile (T*C, Hs, Ws) int16|f32, origins (n, 2) int32 rows (top, left) on the
    device -> (n, C, T, crop, crop) f32 normalised [+ the same windows of ``labels`` (Hs, Ws) f32 -> (n, crop, crop)]
    (process_test / crop_array, dataloader.py:588-669)."""
    TC, Hs, Ws = tile.shape
    C = TC // temporal
    n = origins.shape[0]
    assert C * temporal == TC and mean.numel() == C and std.numel() == C
    assert origins.dtype == torch.int32 and origins.dim() == 2 and origins.shape[1] == 2 and origins.device == tile.device
    dt = {torch.int16: 0, torch.float32: 1}[tile.dtype]
    if out is None:
        out = torch.empty((n, C, temporal, crop, crop), dtype=torch.float32, device=tile.device)
    else:
        assert out.shape == (n, C, temporal, crop, crop) and out.dtype == torch.float32
    lab_out = None
    if labels is not None:
        labels = _f32(labels)
        assert labels.shape == (Hs, Ws)
        lab_out = torch.empty((n, crop, crop), dtype=torch.float32, device=tile.device)
    mult = 1.0 if constant_multiplier is None else float(constant_multiplier)
    work = float(n) * TC * crop * crop * (tile.element_size() + 4) + (float(n) * crop * crop * 8 if labels is not None else 0.0)
    _call("ig_normalize_windows", work, _p(tile), dt, _p(_f32(mean)), _p(_f32(std)), mult, int(constant_multiplier is not None),
          _p(origins.contiguous()), _p(out), _p(labels), _p(lab_out), n, temporal, C, Hs, Ws, crop, _stream())
    return out, lab_out


def chip_stats(x: torch.Tensor, sums: torch.Tensor) -> None:
    """x (B, C, T, H, W) f32; sums (2C,) f64: sums[c] += per-chip mean, sums[C+c] += per-chip biased variance (mode=stats)."""
    B, C = x.shape[0], x.shape[1]
    n = x.numel() // max(B * C, 1)
    assert sums.dtype == torch.float64 and sums.numel() == 2 * C
    _call("ig_chip_stats", float(x.numel()) * 8, _p(_f32(x)), _p(sums), B, C, n, _stream())


def label_hist(labels: torch.Tensor, counts: torch.Tensor, lo: int = -1) -> None:
    """counts (nbins + 1,) int64: counts[v - lo] += #pixels with integer label v; counts[-1] collects everything else."""
    assert counts.dtype == torch.int64
    _call("ig_label_hist", float(labels.numel()) * 4, _p(_f32(labels)), _p(counts), labels.numel(), lo, counts.numel() - 1, _stream())


def patchify(img: torch.Tensor, p: int, out: BT) -> None:
    B, C, T, H, W = img.shape
    _lib.call("ig_patchify", _p(_f32(img)), _p(out.hi), _p(out.lo), B, C, T, H, W, p, _stream())


def cls_rows(x: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, B: int, ntok: int, D: int) -> None:
    _lib.call("ig_cls_rows", _p(x), _p(cls), _p(pos), B, ntok, D, _stream())


def patch_embed_fwd(patches: BT, w: BT, bias, pos, x, batch: int, tpc: int, D: int, K: int) -> None:
    _call("ig_patch_embed_fwd", 2.0 * batch * tpc * D * K, _p(patches.hi), _p(patches.lo), _p(w.hi), _p(w.lo), _p(bias), _p(pos), _p(x), batch, tpc, D, K,
              _stream())


def layernorm_fwd(x, gamma, beta, out: BT, mean, rstd, M: int, D: int, eps: float = 1e-5, feat_T: int = 0, feat_G: int = 0,
                  ntok: int = 0) -> None:
    # algorithmic bytes: fp32 residual stream in, bf16 (hi [+ lo]) out
    _call("ig_layernorm_fwd", float(M) * D * (4 + (4 if out.lo is not None else 2)), _p(x), _p(gamma), _p(beta), _p(out.hi), _p(out.lo),
          _p(mean), _p(rstd), M, D, eps, feat_T, feat_G, ntok, _stream())


def layernorm_bwd(dy: BT, x, mean, rstd, gamma, dx, accumulate: bool, dxb: Optional[BT], dgamma, dbeta, dcol, M: int, D: int,
                  feat_T: int = 0, feat_G: int = 0, ntok: int = 0) -> None:
    nb = 2 if dy.lo is None else 4
    work = float(M) * D * (nb + 4 + 4 + (4 if accumulate else 0) + (nb if dxb else 0))  # dy, x, dx (r)w, bf16 copy of dx
    _call("ig_layernorm_bwd", work, _p(dy.hi), _p(dy.lo), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), int(accumulate),
          _p(dxb.hi) if dxb else None, _p(dxb.lo) if dxb else None, _p(dgamma), _p(dbeta), _p(dcol), M, D, feat_T, feat_G, ntok,
          _stream())


def linear_fwd(x: BT, w: BT, bias, y: BT, M: int, N: int, K: int, act: int = 0, pre: Optional[BT] = None) -> None:
    """y = act(x @ w^T + b).  With act=1 and ``pre`` given, ``pre`` receives gelu'(x @ w^T + b) -- the elementwise factor
    :func:`linear_dgrad` applies in backward (the pre-activation itself is never needed again)."""
    _call("ig_linear_fwd", 2.0 * M * N * K, _p(x.hi), _p(x.lo), _p(w.hi), _p(w.lo), _p(bias), _p(y.hi), _p(y.lo),
              _p(pre.hi) if pre else None, _p(pre.lo) if pre else None, M, N, K, act, _stream())


def linear_residual_fwd(x: BT, w: BT, bias, resid, out, M: int, N: int, K: int) -> None:
    _call("ig_linear_residual_fwd", 2.0 * M * N * K, _p(x.hi), _p(x.lo), _p(w.hi), _p(w.lo), _p(bias), _p(resid), _p(out), M, N, K, _stream())


def linear_dgrad(dy: BT, w: Optional[BT], dx: BT, M: int, N: int, K: int, pre: Optional[BT] = None, colsum=None,
                 wt: Optional[BT] = None) -> None:
    """dx = dy @ w [* pre, the gelu' saved by linear_fwd]; ``colsum`` (fp32 [K]) additionally accumulates the column sums of dx.
    ``wt`` = the same weight stored transposed ([K][N], :func:`transpose_bf16`): the K-contiguous form (``ig_linear_dgrad_wt``)."""
    if wt is not None:
        _call("ig_linear_dgrad", 2.0 * M * N * K, _p(dy.hi), _p(dy.lo), _p(wt.hi), _p(wt.lo), _p(dx.hi), _p(dx.lo),
              _p(pre.hi) if pre else None, _p(pre.lo) if pre else None, _p(colsum), M, N, K, 1 if pre else 0, _stream(),
              entry="ig_linear_dgrad_wt")
        return
    _call("ig_linear_dgrad", 2.0 * M * N * K, _p(dy.hi), _p(dy.lo), _p(w.hi), _p(w.lo), _p(dx.hi), _p(dx.lo),
              _p(pre.hi) if pre else None, _p(pre.lo) if pre else None, _p(colsum), M, N, K, 1 if pre else 0, _stream())


def transpose_bf16(src: BT, dst: BT, R: int, C: int, batch: int = 1, src_stride: int = 0, dst_stride: int = 0) -> None:
    """dst[b] (C, R) = src[b] (R, C)^T for b < batch (strides in elements between consecutive matrices)."""
    _lib.call("ig_transpose_bf16", _p(src.hi), _p(src.lo), _p(dst.hi), _p(dst.lo), R, C, batch, src_stride, dst_stride, _stream())


def linear_wgrad(dy: BT, x: BT, dw, M: int, N: int, K: int) -> None:
    _call("ig_linear_wgrad", 2.0 * M * N * K, _p(dy.hi), _p(dy.lo), _p(x.hi), _p(x.lo), _p(dw), M, N, K, _stream())


class WgradGroup:
    """Prepared argument block of one grouped weight-gradient launch: ``items`` = [(dy, x, dw, N, K), ...] sharing the token
    count M.  The ctypes pointer / size arrays are built ONCE (building them per call costs tens of microseconds of Python --
    at small batches the step is bound by the host's launch rate), the tensors are kept alive by the object."""

    __slots__ = ("items", "M", "work", "_args")

    def __init__(self, items, M: int):
        import ctypes

        n = len(items)
        vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
        split = items[0][0].lo is not None
        arrs = [vp(*[_p(it[0].hi) for it in items]), vp(*[_p(it[0].lo) for it in items]) if split else None,
                vp(*[_p(it[1].hi) for it in items]), vp(*[_p(it[1].lo) for it in items]) if split else None,
                vp(*[_p(it[2]) for it in items]), ip(*[int(it[3]) for it in items]), ip(*[int(it[4]) for it in items])]
        self.items, self.M = list(items), int(M)
        self.work = sum(2.0 * M * it[3] * it[4] for it in items)
        self._args = (n, *[None if a is None else ctypes.cast(a, ctypes.c_void_p) for a in arrs], int(M)), arrs  # arrs: keep-alive

    def launch(self, overwrite: bool = False) -> None:
        """``overwrite``: dW = dy^T x instead of dW += (the first backward of a step: no zeroed dW needed, its old contents not read)."""
        _call("ig_linear_wgrad_group", self.work, *self._args[0], int(overwrite), _stream())


def linear_wgrad_group(items, M: int, overwrite: bool = False) -> None:
    """``items`` = [(dy, x, dw, N, K), ...]: dw_g[N_g][K_g] += dy_g[M][N_g]^T @ x_g[M][K_g] for all g in ONE launch (the weight
    gradients of a Block's linears share the token count; grouped, their output tiles fill the CUs with 2-3 token ranges per
    tile instead of 7-28 and the split-K fold shrinks accordingly).  Hot loops keep a :class:`WgradGroup` instead."""
    WgradGroup(items, M).launch(overwrite)


def attention_fwd(qkv: BT, out: BT, lse, B: int, N: int, H: int, hd: int = 64) -> None:
    _call("ig_attention_fwd", 4.0 * B * H * N * N * hd, _p(qkv.hi), _p(qkv.lo), _p(out.hi), _p(out.lo), _p(lse), B, N, H, hd, _stream())


def attention_bwd(qkv: BT, out: BT, dout: BT, lse, delta, dqkv: BT, B: int, N: int, H: int, hd: int = 64, dbias=None) -> None:
    """``dbias`` (fp32 [3*H*hd], optional) += column sums of dqkv over the tokens: the bias gradient of the fused qkv Linear."""
    _call("ig_attention_bwd", 10.0 * B * H * N * N * hd, _p(qkv.hi), _p(qkv.lo), _p(out.hi), _p(out.lo), _p(dout.hi), _p(dout.lo), _p(lse), _p(delta),
              _p(dqkv.hi), _p(dqkv.lo), _p(dbias), B, N, H, hd, _stream())


def colsum(x: BT, out, M: int, C: int) -> None:
    _call("ig_colsum", float(M) * C * (2 if x.lo is None else 4), _p(x.hi), _p(x.lo), _p(out), M, C, _stream())


def patch_grad_prep(dx, out: BT, dcls, dbias, B: int, ntok: int, D: int) -> None:
    _lib.call("ig_patch_grad_prep", _p(dx), _p(out.hi), _p(out.lo), _p(dcls), _p(dbias), B, ntok, D, _stream())


def convT_fwd(x: BT, w: BT, bias, y: BT, B, H, W, Cin, Cout, seed: int = 0, p: float = 0.0, seed_dev=None) -> None:
    _call("ig_convT_fwd", 2.0 * B * H * W * Cin * Cout * 9, _p(x.hi), _p(x.lo), _p(w.hi), _p(w.lo), _p(bias), _p(y.hi), _p(y.lo), B, H, W, Cin, Cout, seed, _p(seed_dev), p,
              _stream())


def convT_dgrad(dy: BT, w: BT, dx: BT, B, H, W, Cin, Cout) -> None:
    _call("ig_convT_dgrad", 2.0 * B * H * W * Cin * Cout * 9, _p(dy.hi), _p(dy.lo), _p(w.hi), _p(w.lo), _p(dx.hi), _p(dx.lo), B, H, W, Cin, Cout, _stream())


def convT_wgrad(dy: BT, x: BT, dw, B, H, W, Cin, Cout, dbias=None) -> None:
    """dw += per-tap dy^T x; ``dbias`` (fp32 [Cout]) additionally accumulates the column sums of dy (the bias gradient)."""
    _call("ig_convT_wgrad", 2.0 * B * H * W * Cin * Cout * 9, _p(dy.hi), _p(dy.lo), _p(x.hi), _p(x.lo), _p(dw), _p(dbias), B, H, W, Cin, Cout, _stream())


def conv3x3_fwd(x: BT, w: BT, bias, y: BT, B, H, W, Cin, Cout, bn_scale=None, bn_shift=None) -> None:
    """3x3 conv (+bias); with bn_scale/bn_shift also eval-mode BatchNorm + ReLU in the same epilogue."""
    _call("ig_conv3x3_fwd", 2.0 * B * H * W * Cin * Cout * 9, _p(x.hi), _p(x.lo), _p(w.hi), _p(w.lo), _p(bias), _p(bn_scale), _p(bn_shift),
          _p(y.hi), _p(y.lo), B, H, W, Cin, Cout, _stream())


def conv_fwd(x: BT, w: BT, bias, y: BT, B, H, W, Cin, Cout, ks: int = 3, bn_scale=None, bn_shift=None) -> None:
    """nn.Conv2d(kernel_size=ks, padding=1): ks = 3 is :func:`conv3x3_fwd`; 5 / 7 (the 600M head) shrink the map to H + 3 - ks."""
    if ks == 3:
        return conv3x3_fwd(x, w, bias, y, B, H, W, Cin, Cout, bn_scale, bn_shift)
    Ho, Wo = H + 3 - ks, W + 3 - ks
    _call("ig_conv3x3_fwd", 2.0 * B * Ho * Wo * Cin * Cout * ks * ks, _p(x.hi), _p(x.lo), _p(w.hi), _p(w.lo), _p(bias), _p(bn_scale), _p(bn_shift),
          _p(y.hi), _p(y.lo), B, H, W, Cin, Cout, ks, _stream(), entry="ig_convk_fwd")


def conv_dgrad(dy: BT, w: BT, dx: BT, B, H, W, Cin, Cout, ks: int = 3, seed: int = 0, p: float = 0.0, seed_dev=None) -> None:
    if ks == 3:
        return conv3x3_dgrad(dy, w, dx, B, H, W, Cin, Cout, seed, p, seed_dev)
    Ho, Wo = H + 3 - ks, W + 3 - ks
    _call("ig_conv3x3_dgrad", 2.0 * B * Ho * Wo * Cin * Cout * ks * ks, _p(dy.hi), _p(dy.lo), _p(w.hi), _p(w.lo), _p(dx.hi), _p(dx.lo), B, H, W, Cin,
          Cout, ks, seed, _p(seed_dev), p, _stream(), entry="ig_convk_dgrad")


def conv_wgrad(dy: BT, x: BT, dw, B, H, W, Cin, Cout, ks: int = 3, dbias=None) -> None:
    if ks == 3:
        return conv3x3_wgrad(dy, x, dw, B, H, W, Cin, Cout, dbias)
    Ho, Wo = H + 3 - ks, W + 3 - ks
    _call("ig_conv3x3_wgrad", 2.0 * B * Ho * Wo * Cin * Cout * ks * ks, _p(dy.hi), _p(dy.lo), _p(x.hi), _p(x.lo), _p(dw), _p(dbias), B, H, W, Cin, Cout,
          ks, _stream(), entry="ig_convk_wgrad")


def bn_eval_affine(gamma, beta, rmean, rvar, scale, shift, C: int, eps: float = 1e-5) -> None:
    _lib.call("ig_bn_eval_affine", _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(scale), _p(shift), C, eps, _stream())


def conv3x3_dgrad(dy: BT, w: BT, dx: BT, B, H, W, Cin, Cout, seed: int = 0, p: float = 0.0, seed_dev=None) -> None:
    _call("ig_conv3x3_dgrad", 2.0 * B * H * W * Cin * Cout * 9, _p(dy.hi), _p(dy.lo), _p(w.hi), _p(w.lo), _p(dx.hi), _p(dx.lo), B, H, W, Cin, Cout, seed, _p(seed_dev), p,
              _stream())


def conv3x3_wgrad(dy: BT, x: BT, dw, B, H, W, Cin, Cout, dbias=None) -> None:
    """dw += dy^T x_gathered; ``dbias`` (fp32 [Cout]) additionally accumulates the column sums of dy (the bias gradient)."""
    _call("ig_conv3x3_wgrad", 2.0 * B * H * W * Cin * Cout * 9, _p(dy.hi), _p(dy.lo), _p(x.hi), _p(x.lo), _p(dw), _p(dbias), B, H, W, Cin, Cout, _stream())


def bn_relu_fwd(x: BT, gamma, beta, rmean, rvar, y: BT, scale, shift, mean, rstd, sums, M: int, C: int, training: bool,
                update_running: bool, eps: float = 1e-5, momentum: float = 0.1, stats_ready: bool = False) -> None:
    """``stats_ready``: ``sums`` already holds the batch statistics (:func:`conv3x3_fwd_stats` returned True): no statistics pass."""
    nb = 2 if x.lo is None else 4
    # training: statistics pass (read) + apply pass (read + write); eval: one read + write pass
    mode = (2 if stats_ready else 1) if training else 0
    _call("ig_bn_relu_fwd", float(M) * C * nb * (3 if mode == 1 else 2), _p(x.hi), _p(x.lo), _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(y.hi), _p(y.lo), _p(scale), _p(shift),
          _p(mean), _p(rstd), _p(sums), M, C, eps, momentum, mode, int(update_running), _stream())


def bn_relu_bwd(x: BT, dy: BT, scale, shift, mean, rstd, dx: BT, dgamma, dbeta, sums, M: int, C: int) -> None:
    nb = 2 if x.lo is None else 4
    # reduction pass reads x, dy; apply pass reads x, dy and writes dx
    _call("ig_bn_relu_bwd", float(M) * C * nb * 5, _p(x.hi), _p(x.lo), _p(dy.hi), _p(dy.lo), _p(scale), _p(shift), _p(mean), _p(rstd),
          _p(dx.hi), _p(dx.lo), _p(dgamma), _p(dbeta), _p(sums), M, C, _stream())


def classifier_fwd(f: BT, w, bias, logits, B: int, HW: int, C: int, ncls: int, seed: int = 0, p: float = 0.0, seed_dev=None) -> None:
    _call("ig_classifier_fwd", float(B) * HW * (C * (2 if f.lo is None else 4) + ncls * 4), _p(f.hi), _p(f.lo), _p(w), _p(bias), _p(logits), B, HW, C, ncls, seed, _p(seed_dev), p, _stream())


def classifier_bwd(dlogits, f: BT, w, df: BT, dw, db, count, B: int, HW: int, C: int, ncls: int, seed: int = 0, p: float = 0.0,
                   seed_dev=None) -> None:
    _call("ig_classifier_bwd", float(B) * HW * (2 * C * (2 if f.lo is None else 4) + ncls * 4), _p(dlogits), _p(f.hi), _p(f.lo), _p(w),
          _p(df.hi), _p(df.lo), _p(dw), _p(db), _p(count), B, HW, C, ncls, seed, _p(seed_dev), p, _stream())


def bn_stats(x: BT, gamma, beta, rmean, rvar, scale, shift, mean, rstd, sums, M: int, C: int, update_running: bool, eps: float = 1e-5,
             momentum: float = 0.1) -> None:
    """Training-mode BatchNorm statistics only (scale / shift / mean / rstd, running update): the consumer applies them."""
    _call("ig_bn_relu_fwd", float(M) * C * (2 if x.lo is None else 4), _p(x.hi), _p(x.lo), _p(gamma), _p(beta), _p(rmean), _p(rvar), None, None,
          _p(scale), _p(shift), _p(mean), _p(rstd), _p(sums), M, C, eps, momentum, 1, int(update_running), _stream())


def conv3x3_fwd_stats(x: BT, w: BT, bias, y: BT, sums, B, H, W, Cin, Cout) -> bool:
    """nn.Conv2d(k=3, padding=1) in front of a training-mode BatchNorm; True when the kernel also left the per-channel sum / sum of squares
    of its outputs in ``sums`` (f64 [2 Cout]): the BatchNorm then needs :func:`bn_finalize` only, not a statistics pass."""
    import ctypes

    fused = ctypes.c_int(0)
    _call("ig_conv3x3_fwd", 2.0 * B * H * W * Cin * Cout * 9, _p(x.hi), _p(x.lo), _p(w.hi), _p(w.lo), _p(bias), _p(y.hi), _p(y.lo), _p(sums),
          ctypes.cast(ctypes.byref(fused), ctypes.c_void_p), B, H, W, Cin, Cout, _stream(), entry="ig_conv3x3_fwd_stats")
    return bool(fused.value)


def conv3x3_cls_fwd(x: BT, w: BT, bias, bn_scale, bn_shift, y: Optional[BT], cls_w, cls_b, logits, B, H, W, C, ncls) -> bool:
    """Inference tail in one kernel: the last 3 x 3 convolution (+ eval-mode BatchNorm + ReLU) and the 1 x 1 classifier.  True when it ran
    (direct 48-channel kernel, <= 2 classes, plain bf16); False: nothing was computed, run :func:`conv3x3_fwd` + :func:`classifier_fwd`.
    ``y`` may be None: the activation is then not stored."""
    import ctypes

    if x.lo is not None:
        return False
    fused = ctypes.c_int(0)
    _call("ig_conv3x3_fwd", 2.0 * B * H * W * C * C * 9, _p(x.hi), None, _p(w.hi), None, _p(bias), _p(bn_scale), _p(bn_shift), _p(y.hi) if y is not None else None,
          _p(cls_w), _p(cls_b), _p(logits), ctypes.cast(ctypes.byref(fused), ctypes.c_void_p), B, H, W, C, C, ncls, _stream(), entry="ig_conv3x3_cls_fwd")
    return bool(fused.value)


def bn_finalize(sums, gamma, beta, rmean, rvar, scale, shift, mean, rstd, M: int, C: int, update_running: bool, eps: float = 1e-5,
                momentum: float = 0.1) -> None:
    _lib.call("ig_bn_finalize", _p(sums), _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(scale), _p(shift), _p(mean), _p(rstd), M, C, eps, momentum,
              int(update_running), _stream())


def classifier_bn_fwd(x: BT, scale, shift, w, bias, logits, B: int, HW: int, C: int, ncls: int, seed: int = 0, p: float = 0.0, seed_dev=None) -> None:
    _call("ig_classifier_bn_fwd", float(B) * HW * (C * (2 if x.lo is None else 4) + ncls * 4), _p(x.hi), _p(x.lo), _p(scale), _p(shift), _p(w),
          _p(bias), _p(logits), B, HW, C, ncls, seed, _p(seed_dev), p, _stream())


def classifier_bn_bwd(dlogits, x: BT, scale, shift, mean, rstd, w, dx: BT, dw, db, dgamma, dbeta, sums, count, B: int, HW: int, C: int,
                      ncls: int, seed: int = 0, p: float = 0.0, seed_dev=None) -> None:
    # two passes over x (+ dlogits), one write of dx
    _call("ig_classifier_bn_bwd", float(B) * HW * (3 * C * (2 if x.lo is None else 4) + 2 * ncls * 4), _p(dlogits), _p(x.hi), _p(x.lo), _p(scale),
          _p(shift), _p(mean), _p(rstd), _p(w), _p(dx.hi), _p(dx.lo), _p(dw), _p(db), _p(dgamma), _p(dbeta), _p(sums), _p(count), B, HW, C, ncls,
          seed, _p(seed_dev), p, _stream())


_LABEL_DT = {torch.int64: 0, torch.int32: 1, torch.float32: 2}


def ce_loss(logits, labels, class_weights, ignore_index: int, stats, dlogits=None, preds=None, preds_i8=None, confusion=None) -> None:
    B, ncls = logits.shape[0], logits.shape[1]
    HW = logits.numel() // (B * ncls)
    work = float(B) * HW * (ncls * 4 + labels.element_size() + (ncls * 4 if dlogits is not None else 0) + (8 if preds is not None else 0)
                            + (1 if preds_i8 is not None else 0))
    _call("ig_ce_loss", work, _p(_f32(logits)), _p(labels), _LABEL_DT[labels.dtype], _p(class_weights), int(ignore_index), _p(stats),
          _p(dlogits), _p(preds), _p(preds_i8), _p(confusion), B, HW, ncls, _stream())


def kd_loss(student_logits, teacher_logits, labels, ignore_index: Optional[int], kl_sum, dlogits=None) -> None:
    """KLDivLoss(batchmean) numerator over the valid pixels into ``kl_sum`` (f64 [1]); ``dlogits`` += softmax(s) - softmax(t)."""
    B, ncls = student_logits.shape[0], student_logits.shape[1]
    HW = student_logits.numel() // (B * ncls)
    assert teacher_logits.shape == student_logits.shape and kl_sum.dtype == torch.float64
    ign = -(2**62) if ignore_index is None else int(ignore_index)
    _call("ig_kd_loss", float(B) * HW * ncls * 12, _p(_f32(student_logits)), _p(_f32(teacher_logits)), _p(labels), _LABEL_DT[labels.dtype], ign,
          _p(kl_sum), _p(dlogits), B, HW, ncls, _stream())


def mse_loss(pred, labels, ignore_index: float, use_log_scale: bool, stats, dpred=None, msums=None, ee_bias: float = 0.05,
             ee_coef: float = 0.15, include_ee: bool = False) -> None:
    """Masked MSE of the regression head + streaming regression-metric sums (regression.py:141-191, metrics.py:330-352)."""
    n = pred.numel()
    assert labels.numel() == n and stats.dtype == torch.float64
    work = float(n) * (8 + (4 if dpred is not None else 0))
    _call("ig_mse_loss", work, _p(_f32(pred)), _p(_f32(labels)), float(ignore_index), int(use_log_scale), _p(stats), _p(dpred), _p(msums),
          float(ee_bias), float(ee_coef), int(include_ee), n, _stream())


def kd_mse_loss(pred, teacher, labels, ignore_index: float, use_log_scale: bool, total, dpred=None) -> None:
    """Regression distillation term: ``total`` (f64 [1]) += sum over valid pixels of (pred - teacher')^2; ``dpred`` += 2 (pred - teacher')
    (regression.py:477-534)."""
    n = pred.numel()
    assert teacher.numel() == n and labels.numel() == n and total.dtype == torch.float64
    _call("ig_kd_mse_loss", float(n) * (12 + (8 if dpred is not None else 0)), _p(_f32(pred)), _p(_f32(teacher)), _p(_f32(labels)),
          float(ignore_index), int(use_log_scale), _p(total), _p(dpred), n, _stream())


def auc_update(logits, labels, ignore_index: Optional[int], hist, nbins: int, min_score: float = 0.0, max_score: float = 1.0) -> None:
    """RunningAUC histograms of softmax(logits): hist int64 [2, ncls, nbins] (0 positives, 1 negatives of each class)."""
    B, ncls = logits.shape[0], logits.shape[1]
    HW = logits.numel() // (B * ncls)
    assert hist.dtype == torch.int64 and hist.numel() == 2 * ncls * nbins
    ign = -(2**62) if ignore_index is None else int(ignore_index)
    _call("ig_auc_update", float(B) * HW * (ncls * 4 + labels.element_size()), _p(_f32(logits)), _p(labels), _LABEL_DT[labels.dtype], ign,
          _p(hist), B, HW, ncls, nbins, float(min_score), float(max_score), _stream())


def softmax_prob(logits, cls: int = 1, out=None):
    """softmax(logits, dim=1)[:, cls] -> (B, H, W) f32 (predict_step, segmentation.py:202-213)."""
    B, ncls = logits.shape[0], logits.shape[1]
    HW = logits.numel() // (B * ncls)
    if out is None:
        out = torch.empty((B,) + tuple(logits.shape[2:]), dtype=torch.float32, device=logits.device)
    _call("ig_softmax_prob", float(B) * HW * (ncls * 4 + 4), _p(_f32(logits)), _p(out), B, HW, ncls, int(cls), _stream())
    return out


def argmax_i8(logits, out=None):
    B, ncls = logits.shape[0], logits.shape[1]
    HW = logits.numel() // (B * ncls)
    if out is None:
        out = torch.empty((B,) + tuple(logits.shape[2:]), dtype=torch.int8, device=logits.device)
    _call("ig_argmax_i8", float(B) * HW * (ncls * 4 + 1), _p(_f32(logits)), _p(out), B, HW, ncls, _stream())
    return out


def confusion_update(y_true, y_pred, confusion, k: int, ignore_index: Optional[int]) -> None:
    assert y_true.dtype == torch.int64 and y_pred.dtype == torch.int64 and confusion.dtype == torch.int64
    _lib.call("ig_confusion_update", _p(y_true), _p(y_pred), _p(confusion), y_true.numel(), k,
              0 if ignore_index is None else int(ignore_index), int(ignore_index is not None), _stream())


def adamw_advance(hyper) -> None:
    _lib.call("ig_adamw_advance", _p(hyper), _stream())


def adamw_step(p, g, m, v, shadow: Optional[BT], hyper, n: int) -> None:
    # 28 B/param fp32 state (read p, g, m, v; write p, m, v) + the refreshed bf16 shadow
    work = float(n) * (28 + (0 if not shadow else 2 if shadow.lo is None else 4))
    _call("ig_adamw_step", work, _p(p), _p(g), _p(m), _p(v), _p(shadow.hi) if shadow else None,
          _p(shadow.lo) if shadow and shadow.lo is not None else None, _p(hyper), n, _stream())
