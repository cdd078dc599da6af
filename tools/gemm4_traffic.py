#!/usr/bin/env python3
"""The plain kind of gemm4 (qkv forward and the three plain data gradients share the kernel name gemm4_kernel<0,0,false>) launched shape by
shape, 6 launches each in a fixed order, for the FETCH_SIZE / WRITE_SIZE passes of rocprofv3: tools/gemm4_traffic.sh groups the counter rows
by dispatch order.  usage: python tools/gemm4_traffic.py [M]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instageo-e2e-geospatial-ml_amd"))
import torch  # noqa: E402

from instageo_amd import ops  # noqa: E402
from instageo_amd.ops import BT  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 432 * 197
D = 768
dev = "cuda"
SHAPES = [("qkv fwd", 3 * D, D), ("d_qkv", D, 3 * D), ("d_proj", D, D), ("d_fc1", D, 4 * D)]  # (name, N, K): out[M][N] = x[M][K] w[N][K]^T
for name, N, K in SHAPES:
    xs = [BT.from_float(torch.randn(M, K, device=dev), False) for _ in range(3)]  # rotate inputs: nothing is L2 / MALL resident from the last launch
    w = BT.from_float(torch.randn(N, K, device=dev) * K**-0.5, False)
    y = BT.empty((M, N), False, dev)
    for i in range(6):
        ops.linear_fwd(xs[i % 3], w, None, y, M, N, K, act=0, pre=None)
    assert ops.last_kernel().startswith("gemm4_kernel<0,0,false>"), ops.last_kernel()
    torch.cuda.synchronize()
print("launched", [s[0] for s in SHAPES], "x 6 at M =", M)
