#!/usr/bin/env python3
"""Micro-benchmark of the LayerNorm backward / column-sum kernels (HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instageo-e2e-geospatial-ml_amd"))
import torch
from instageo_amd import ops
from instageo_amd.ops import BT
dev = "cuda"
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
M, D = 21276, 768
x = torch.randn(M, D, device=dev); g = torch.ones(D, device=dev); bta = torch.zeros(D, device=dev)
out = BT.empty((M, D), False, dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
ops.layernorm_fwd(x, g, bta, out, mean, rstd, M, D)
dy = BT(torch.randn(M, D, device=dev).bfloat16()); dx = torch.zeros(M, D, device=dev); dxb = BT.empty((M, D), False, dev)
dg, db, dc = (torch.zeros(D, device=dev) for _ in range(3))
print("ln fwd us", timeit(lambda: ops.layernorm_fwd(x, g, bta, out, mean, rstd, M, D)))
print("ln bwd full us", timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dx, True, dxb, dg, db, dc, M, D)))
print("ln bwd no column outputs us", timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dx, True, dxb, None, None, None, M, D)))
print("ln bwd no accumulate, no dxb us", timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dx, False, None, None, None, None, M, D)))
big = BT(torch.randn(M, 3072, device=dev).bfloat16()); o = torch.zeros(3072, device=dev)
print("colsum M x 3072 us", timeit(lambda: ops.colsum(big, o, M, 3072)))
q = BT(torch.randn(M, 2304, device=dev).bfloat16()); o2 = torch.zeros(2304, device=dev)
print("colsum M x 2304 us", timeit(lambda: ops.colsum(q, o2, M, 2304)))
