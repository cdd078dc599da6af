#!/usr/bin/env python3
"""Micro-benchmark of the C-ABI GEMM entry points (HIP events, random data).  Usage: python tools/gemm_bench.py [shapes...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instageo-e2e-geospatial-ml_amd"))
import torch

from instageo_amd import ops
from instageo_amd.ops import BT

dev = "cuda"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


def rnd(*s):
    return BT(torch.randn(*s, device=dev).bfloat16())


shapes = [(12608, 2304, 768), (12608, 3072, 768), (12608, 768, 768), (12608, 768, 3072), (8192, 8192, 8192), (4096, 4096, 4096),
          (10752, 2304, 768), (10752, 768, 3072)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]

for M, N, K in shapes:
    x, w, y = rnd(M, K), rnd(N, K), BT(torch.empty(M, N, device=dev, dtype=torch.bfloat16))
    bias = torch.zeros(N, device=dev)
    fl = 2.0 * M * N * K
    t_fwd = timeit(lambda: ops.linear_fwd(x, w, bias, y, M, N, K))
    t_gelu = timeit(lambda: ops.linear_fwd(x, w, bias, y, M, N, K, act=1))
    res = torch.zeros(M, N, device=dev)
    t_res = timeit(lambda: ops.linear_residual_fwd(x, w, bias, res, res, M, N, K))
    dy, dx = rnd(M, N), BT(torch.empty(M, K, device=dev, dtype=torch.bfloat16))
    t_dg = timeit(lambda: ops.linear_dgrad(dy, w, dx, M, N, K))
    pre = rnd(M, K)
    t_dgg = timeit(lambda: ops.linear_dgrad(dy, w, dx, M, N, K, pre=pre))
    cs = torch.zeros(K, device=dev)
    t_dgc = timeit(lambda: ops.linear_dgrad(dy, w, dx, M, N, K, colsum=cs))
    dw = torch.zeros(N, K, device=dev)
    t_wg = timeit(lambda: ops.linear_wgrad(dy, x, dw, M, N, K))
    print(f"M{M} N{N} K{K}: fwd {fl/t_fwd/1e12:6.0f}  fwd+gelu {fl/t_gelu/1e12:6.0f}  resid {fl/t_res/1e12:6.0f}  dgrad {fl/t_dg/1e12:6.0f} (+gelu' {fl/t_dgg/1e12:4.0f}, +colsum {fl/t_dgc/1e12:4.0f})  wgrad {fl/t_wg/1e12:6.0f} TFLOP/s   (fwd {t_fwd*1e6:.0f} us)")
