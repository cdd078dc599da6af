"""Operand-format study on the CPU oracle (can a ONE-pass MFMA mode meet the 1e-3 bar?).

Every contraction of the path (linears, patch embedding, QK^T, PV, ConvTranspose / Conv2d) is evaluated with BOTH operands rounded
to a storage format, fp32 accumulation, exactly where the HIP path rounds them (GEMM / convolution operands are stored in the
format; the residual stream, LayerNorm statistics, softmax, BatchNorm and the classifier weights stay fp32), and the eval logits /
train-mode loss / mIoU are compared with the unrounded fp32 oracle on the six parity cases of oracle/cases.py:

    bf16    8-bit mantissa, one MFMA pass          (the benchmarked mode)
    fp16    11-bit mantissa, one MFMA pass         (same MFMA rate as bf16)
    bf16x2  A = hi + lo (16 bits), B = hi (8 bits): two passes (hi*hi + lo*hi)
    bf16x3  both operands hi + lo, the lo*lo term dropped: three passes (the parity mode)

Run in the build container (CPU): python tools/precision_study.py [--cases tiny_t1_c2,...]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import prithvi_oracle as O  # noqa: E402
from oracle.cases import CASES, EVAL_ONLY, case_config, class_weights_for, make_inputs  # noqa: E402


def rnd(x, fmt):
    if fmt == "bf16":
        return x.to(torch.bfloat16).float()
    if fmt == "fp16":
        return x.to(torch.float16).float()
    raise ValueError(fmt)


def split(x):
    hi = x.to(torch.bfloat16).float()
    return hi, (x - hi).to(torch.bfloat16).float()


class Emu:
    """Patches the functional ops the oracle calls so that contraction operands are rounded to ``fmt``."""

    def __init__(self, fmt):
        self.fmt = fmt
        self.orig = {k: getattr(F, k) for k in ("linear", "conv2d", "conv_transpose2d", "scaled_dot_product_attention", "conv3d")}

    def contract(self, op, a, w, *args, **kw):
        fmt = self.fmt
        if fmt in ("bf16", "fp16"):
            return op(rnd(a, fmt), rnd(w, fmt), *args, **kw)
        ah, al = split(a)
        wh, wl = split(w)
        bias = args[0] if args else kw.pop("bias", None)
        rest = args[1:]
        out = op(ah, wh, bias, *rest, **kw) + op(al, wh, None, *rest, **kw)
        if fmt == "bf16x3":
            out = out + op(ah, wl, None, *rest, **kw)
        return out

    def __enter__(self):
        e = self
        o = self.orig

        def linear(x, w, b=None):
            if w.shape[0] <= 16:  # the 1x1 classifier runs on fp32 weights in the HIP path
                return o["linear"](x, w, b)
            return e.contract(o["linear"], x, w, b)

        def conv2d(x, w, b=None, *a, **k):
            if w.shape[-1] == 1:  # classifier (fp32 weights, stored-format features)
                return o["conv2d"](e.store(x), w, b, *a, **k)
            return e.store(e.contract(o["conv2d"], x, w, b, *a, **k))

        def convT(x, w, b=None, *a, **k):
            return e.store(e.contract(o["conv_transpose2d"], x, w, b, *a, **k))

        def conv3d(x, w, b=None, *a, **k):
            return e.contract(o["conv3d"], x, w, b, *a, **k)

        def sdpa(q, k, v, *a, **kw):
            q, k, v = e.store(q), e.store(k), e.store(v)
            s = (q @ k.transpose(-1, -2)) * q.shape[-1] ** -0.5 if e.fmt in ("bf16", "fp16") else None
            if s is None:
                qh, ql = split(q)
                kh, kl = split(k)
                s = qh @ kh.transpose(-1, -2) + ql @ kh.transpose(-1, -2)
                if e.fmt == "bf16x3":
                    s = s + qh @ kl.transpose(-1, -2)
                s = s * q.shape[-1] ** -0.5
            p = torch.softmax(s, -1)
            if e.fmt in ("bf16", "fp16"):
                return e.store(rnd(p, e.fmt) @ v)
            ph, pl = split(p)
            vh, vl = split(v)
            out = ph @ vh + pl @ vh
            if e.fmt == "bf16x3":
                out = out + ph @ vl
            return e.store(out)

        F.linear, F.conv2d, F.conv_transpose2d, F.scaled_dot_product_attention, F.conv3d = linear, conv2d, convT, sdpa, conv3d
        return self

    def store(self, x):
        """Activations that the HIP path keeps in the operand format between kernels (qkv, attention output, head feature maps)."""
        if self.fmt in ("bf16", "fp16"):
            return rnd(x, self.fmt)
        hi, lo = split(x)
        return hi + lo

    def __exit__(self, *a):
        for k, v in self.orig.items():
            setattr(F, k, v)


def run_case(name, fmts):
    variant, T, ncls, B, depth = CASES[name]
    cfg = case_config(name)
    sd = O.make_state_dict(cfg, seed=1042)
    img, lab = make_inputs(name, cfg, B)
    cw = class_weights_for(ncls)
    rows = []
    with torch.no_grad():
        ref = O.prithvi_seg_forward(cfg, sd, img, training=False)
        ref_t = None if name in EVAL_ONLY else O.prithvi_seg_forward(cfg, sd, img, training=True)
        for fmt in fmts:
            with Emu(fmt):
                got = O.prithvi_seg_forward(cfg, sd, img, training=False)
                got_t = None if ref_t is None else O.prithvi_seg_forward(cfg, sd, img, training=True)
            d = (got - ref).abs()
            row = {"case": name, "fmt": fmt, "max": d.max().item(), "mean": d.mean().item(),
                   "argmax_agree": (got.argmax(1) == ref.argmax(1)).float().mean().item()}
            if ref_t is not None:
                l_ref, l_got = O.seg_loss(ref_t, lab, cw, -1).item(), O.seg_loss(got_t, lab, cw, -1).item()
                m_ref = O.confusion_metrics(O.confusion_matrix(lab.numpy(), ref_t.argmax(1).numpy(), ncls, -1))["jaccard"]
                m_got = O.confusion_metrics(O.confusion_matrix(lab.numpy(), got_t.argmax(1).numpy(), ncls, -1))["jaccard"]
                row.update({"dloss": abs(l_got - l_ref), "dmiou": abs(m_got - m_ref), "train_max": (got_t - ref_t).abs().max().item()})
            rows.append(row)
            print("  ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in row.items()), flush=True)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="tiny_t1_c2,tiny_t3_c13,v1_100_t1_c2,v1_100_t3_c13,v2_300_t1_c2,v2_600_t1_c2")
    ap.add_argument("--fmts", default="bf16,fp16,bf16x2,bf16x3")
    a = ap.parse_args()
    torch.set_num_threads(8)
    for name in a.cases.split(","):
        run_case(name, a.fmts.split(","))


if __name__ == "__main__":
    main()
