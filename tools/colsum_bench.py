#!/usr/bin/env python3
"""Micro-benchmark of ig_colsum over the shapes one training step uses (HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instageo-e2e-geospatial-ml_amd"))
import torch
from instageo_amd import ops
from instageo_amd.ops import BT
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M, C in [(21276, 2304), (21276, 3072), (84672, 384), (338688, 192), (1354752, 96), (5419008, 48)]:
    x = BT(torch.randn(M, C, device=dev).bfloat16()); o = torch.zeros(C, device=dev)
    us = timeit(lambda: ops.colsum(x, o, M, C))
    print(f"colsum {M} x {C}: {us:8.1f} us  {M*C*2/us/1e6:6.2f} TB/s  rpb={os.environ.get('IG_COLSUM_RPB','default')}")
