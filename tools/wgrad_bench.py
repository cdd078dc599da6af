#!/usr/bin/env python3
"""Weight gradients of one Block: four ig_linear_wgrad launches on the 8-phase engine vs ONE grouped launch (ig_linear_wgrad_group)
on the 8-wave kernel (IG_GEMM4=0) and on the 4-wave generated-assembly kernel (gemm4w).  Interleaved rounds, random data, HIP events.
Usage: python tools/wgrad_bench.py [M] [D] [--x3]"""
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


def timeit(fn, n=10):
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


shapes = [("fc2", D, 4 * D), ("fc1", 4 * D, D), ("proj", D, D), ("qkv", 3 * D, D)]
items = [(rnd(M, N), rnd(M, K), torch.zeros(N, K, device=dev), N, K) for _, N, K in shapes]
fl = sum(2.0 * M * N * K for _, N, K in shapes) * (3 if split else 1)


def separate():
    for dy, x, dw, N, K in items:
        ops.linear_wgrad(dy, x, dw, M, N, K)


def grouped():
    ops.linear_wgrad_group(items, M)


def pairs():
    ops.linear_wgrad_group(items[:2], M)
    ops.linear_wgrad_group(items[2:], M)


if "--group-only" in sys.argv:  # profiling passes: a few grouped launches only
    for _ in range(6):
        grouped()
    torch.cuda.synchronize()
    sys.exit(0)
variants = [("v8w x4", {"IG_WGRAD8": "1", "IG_GEMM4": "0"}, separate), ("v8w grouped", {"IG_WGRAD8": "1", "IG_GEMM4": "0"}, grouped),
            ("v4w grouped", {"IG_WGRAD8": "1", "IG_GEMM4": "1"}, grouped)]
res = {v: [] for v, _, _ in variants}
for r in range(5):
    for v, env, fn in variants:
        os.environ.update(env)
        if r == 0:
            for _ in range(2):
                fn()
        res[v].append(timeit(fn))
print(f"M={M} D={D} mode={'bf16x3' if split else 'bf16'}: " + "   ".join(
    f"{v} {statistics.median(t):7.1f} us ({fl / statistics.median(t) / 1e6:5.0f} TF/s)" for v, t in res.items()))
os.environ.pop("IG_WGRAD8", None)
os.environ.pop("IG_GEMM4", None)
for (name, N, K), it in zip(shapes, items):
    t = statistics.median(timeit(lambda it=it: ops.linear_wgrad(it[0], it[1], it[2], M, it[3], it[4])) for _ in range(3))
    print(f"  v8w {name:5s} N={N:5d} K={K:5d}: {t:7.1f} us ({2.0 * M * N * K * (3 if split else 1) / t / 1e6:5.0f} TF/s)")
