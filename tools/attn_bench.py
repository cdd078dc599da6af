#!/usr/bin/env python3
"""Micro-benchmark of the attention entry points (HIP events).  IG_ATTN_CFG / IG_ATTN_MAXW select the geometry."""
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
for B, N, H in [(108, 197, 12), (36, 589, 12), (54, 197, 16)]:
    qkv = BT(torch.randn(B, N, 3 * H * 64, device=dev).bfloat16())
    out = BT.empty((B, N, H * 64), False, dev); lse = torch.empty(B * H * N, device=dev)
    dout = BT(torch.randn(B, N, H * 64, device=dev).bfloat16()); dqkv = BT.empty((B, N, 3 * H * 64), False, dev)
    delta = torch.empty(B * H * N, device=dev)
    os.environ["IG_ATTN2"] = "0"
    tf0 = timeit(lambda: ops.attention_fwd(qkv, out, lse, B, N, H))
    os.environ["IG_ATTN2"] = "1"
    tf = timeit(lambda: ops.attention_fwd(qkv, out, lse, B, N, H))
    os.environ["IG_ATTN2"] = "2"
    tb0 = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H))
    os.environ["IG_ATTN2"] = "1"
    os.environ["IG_ATTN2_DQLB"] = "4"
    os.environ["IG_ATTN2_FUSED"] = "1"
    tbf = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H))
    os.environ["IG_ATTN2_FUSED"] = "0"
    tb4 = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H))
    os.environ["IG_ATTN2_DQLB"] = "2"
    tb = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H))
    fl = 4.0 * B * H * N * N * 64
    print(f"B{B} N{N} H{H}: fwd gen1 {tf0:7.1f} us ({fl/tf0/1e6:5.0f} TF)  gen2 {tf:7.1f} us ({fl/tf/1e6:5.0f} TF)  bwd gen1 {tb0:7.1f} us  gen2(lb4) {tb4:7.1f} us  gen2 {tb:7.1f} us ({2.5*fl/tb/1e6:5.0f} TF)  fused {tbf:7.1f} us ({2.5*fl/tbf/1e6:5.0f} TF)")
