#!/usr/bin/env python3
"""A/B of the forward linear GEMM engines in ONE process (interleaved rounds, random data, HIP events):
the 8-phase engine (gemm8.hip) against the 4-wave engine (gemm4.hip; default) or, with --old, the round-1 engines; --x3: the split mode's
three-pass against its paired form.  Usage: python tools/gemm8_bench.py [M] [D] [--v4 | --old | --x3]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instageo-e2e-geospatial-ml_amd"))
import torch

from instageo_amd import ops
from instageo_amd.ops import BT

dev = "cuda"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
M = int(args[0]) if args else 108 * 197
D = int(args[1]) if len(args) > 1 else 768
split = "--x3" in sys.argv


def timeit(fn, n=20):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


def rnd(*s):
    return BT.from_float(torch.randn(*s, device=dev), split)


cases = []
x = rnd(M, D)
x4 = rnd(M, 4 * D)
for name, N, K, kind in [("qkv", 3 * D, D, "plain"), ("fc1 (gelu, infer)", 4 * D, D, "gelu"), ("fc1 (gelu + gelu', train)", 4 * D, D, "gelu_train"),
                         ("proj (+resid)", D, D, "resid"), ("fc2 (+resid)", D, 4 * D, "resid")]:
    w = BT.from_float(torch.randn(N, K, device=dev) * K**-0.5, split)
    bias = torch.randn(N, device=dev)
    xin = x4 if K == 4 * D else x
    if kind == "resid":
        res = torch.randn(M, N, device=dev)
        out = torch.empty_like(res)
        fn = (lambda xin=xin, w=w, bias=bias, res=res, out=out, N=N, K=K: ops.linear_residual_fwd(xin, w, bias, res, out, M, N, K))
    else:
        y = BT.empty((M, N), split, dev)
        pre = BT.empty((M, N), split, dev) if kind == "gelu_train" else None
        act = 0 if kind == "plain" else 1
        fn = (lambda xin=xin, w=w, bias=bias, y=y, pre=pre, act=act, N=N, K=K: ops.linear_fwd(xin, w, bias, y, M, N, K, act=act, pre=pre))
    cases.append((name, N, K, fn))

# data gradients through the transposed weight copy: d_fc2 (x gelu' + fused fc1-bias column sums) and the plain d_fc1
dy_s, dy_l = rnd(M, D), rnd(M, 4 * D)
wt_fc2 = BT.from_float(torch.randn(4 * D, D, device=dev) * D**-0.5, split)   # (K, N) = fc2.weight^T
wt_fc1 = BT.from_float(torch.randn(D, 4 * D, device=dev) * D**-0.5, split)   # fc1.weight^T
w_fc2 = BT(wt_fc2.hi.t().contiguous(), None if wt_fc2.lo is None else wt_fc2.lo.t().contiguous())
w_fc1 = BT(wt_fc1.hi.t().contiguous(), None if wt_fc1.lo is None else wt_fc1.lo.t().contiguous())
dh, dxs = BT.empty((M, 4 * D), split, dev), BT.empty((M, D), split, dev)
pre = rnd(M, 4 * D)
cs = torch.zeros(4 * D, device=dev)


def d_fc2():
    if os.environ.get("IG_GEMM8") == "0":
        ops.linear_dgrad(dy_s, w_fc2, dh, M, D, 4 * D, pre=pre, colsum=cs)
    else:
        ops.linear_dgrad(dy_s, None, dh, M, D, 4 * D, pre=pre, colsum=cs, wt=wt_fc2)


def d_fc1():
    if os.environ.get("IG_GEMM8") == "0":
        ops.linear_dgrad(dy_l, w_fc1, dxs, M, 4 * D, D)
    else:
        ops.linear_dgrad(dy_l, None, dxs, M, 4 * D, D, wt=wt_fc1)


cases.append(("d_fc2 (x gelu', +colsum)", 4 * D, D, d_fc2))
cases.append(("d_fc1 (plain dgrad)", D, 4 * D, d_fc1))
print(f"M={M} D={D} mode={'bf16x3' if split else 'bf16'}")
if "--v8-only" in sys.argv:  # profiling passes: a few launches of each case on the 8-phase engine only
    os.environ["IG_GEMM8"] = "1"
    for name, N, K, fn in cases:
        for _ in range(6):
            fn()
    torch.cuda.synchronize()
    sys.exit(0)
VARIANTS = [("gemm8", {"IG_GEMM8": "1", "IG_GEMM4": "0"}), ("gemm4", {"IG_GEMM8": "1", "IG_GEMM4": "2"})]
if "--old" in sys.argv:
    VARIANTS = [("old", {"IG_GEMM8": "0"})] + VARIANTS[:1]
if split:  # the split mode: three passes over the operand pairs against the paired K-tiles (hi | lo in one LDS row)
    VARIANTS = [("gemm8 3-pass", {"IG_GEMM8": "1", "IG_G8_PAIR": "0"}), ("gemm8 paired", {"IG_GEMM8": "1", "IG_G8_PAIR": "1", "IG_GEMM4": "0"}),
                ("gemm4 paired", {"IG_GEMM8": "1", "IG_G8_PAIR": "1", "IG_GEMM4": "2"})]
KEYS = ("IG_GEMM8", "IG_G8_PAIR", "IG_GEMM4")
for ci, (name, N, K, fn) in enumerate(cases):
    variants = VARIANTS
    res = {v: [] for v, _ in variants}
    for rnd_i in range(5):
        for v, env in variants:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            if rnd_i == 0:
                for _ in range(3):
                    fn()
            res[v].append(timeit(fn))
    fl = 2.0 * M * N * K * (3 if split else 1)
    line = f"{name:28s} N={N:5d} K={K:5d}: " + "   ".join(f"{v} {statistics.median(t):7.1f} us ({fl/statistics.median(t)/1e6:5.0f} TF/s)" for v, t in res.items())
    print(line)
for k in KEYS:
    os.environ.pop(k, None)
