#!/usr/bin/env python3
"""Phase timing of gemm8_kernel around one tile transition, from a -DIG_G8_PROF build (csrc: `make prof`):

    IG_HIP_LIB=instageo-e2e-geospatial-ml_amd/instageo_amd/libinstageo_hip_g8prof2.so python tools/gemm8_phase_prof.py [M] [case]

case: qkv (default) | fc1 | proj | fc2; --x3: the split precision mode.  Prints, per wave of workgroup 0, s_memtime deltas (shader cycles) of: the tile's epilogue,
the first iteration (two K-tiles) of the next tile, and a plain mid-tile iteration for comparison.
"""
import ctypes
import os
import sys

os.environ["IG_GEMM4"] = "0"  # the stamps live in gemm8.hip: keep the 4-wave engine (which takes the plain / residual kinds by default) out of the way

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "instageo-e2e-geospatial-ml_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from instageo_amd import _lib, ops  # noqa: E402
from instageo_amd.ops import BT  # noqa: E402

split = "--x3" in sys.argv  # the split precision mode (IG_G8_PAIR=0: its three-pass form)
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
M = int(argv[0]) if argv else 42552
case = argv[1] if len(argv) > 1 else "qkv"
D = 768
dev = "cuda"
N, K, kind = {"qkv": (3 * D, D, "plain"), "fc1": (4 * D, D, "gelu_train"), "proj": (D, D, "resid"), "fc2": (D, 4 * D, "resid")}[case]
x = BT.from_float(torch.randn(M, K, device=dev), split)
w = BT.from_float(torch.randn(N, K, device=dev) * K**-0.5, split)
bias = torch.randn(N, device=dev)
if kind == "resid":
    res = torch.randn(M, N, device=dev)
    out = torch.empty_like(res)
    fn = lambda: ops.linear_residual_fwd(x, w, bias, res, out, M, N, K)  # noqa: E731
else:
    y = BT.empty((M, N), split, dev)
    pre = BT.empty((M, N), split, dev) if kind == "gelu_train" else None
    fn = lambda: ops.linear_fwd(x, w, bias, y, M, N, K, act=0 if kind == "plain" else 1, pre=pre)  # noqa: E731
for _ in range(200):  # clocks settle under load
    fn()
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 64))()
lib.ig_debug_g8prof.argtypes = [ctypes.c_void_p]
assert lib.ig_debug_g8prof(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(8, 64).astype(np.int64)
print(f"{case}: M={M} N={N} K={K}; kernel {ops.last_kernel()}")
print("wave:                              " + "".join(f"{w_:8d}" for w_ in range(8)))


def row(label, a, b):
    print(f"{label:35s}" + "".join(f"{int(t[w_, b] - t[w_, a]):8d}" for w_ in range(8)))


row("epilogue: re-align barrier", 32, 33)
row("epilogue: body (stores issued)", 33, 34)
if t[0, 35] > 0:
    row("epilogue: stores acknowledged", 34, 35)
for base, name in ((0, "1st iteration after the epilogue"), (16, "2nd iteration (plain)")):
    print(f"-- {name}")
    row("  stagger + R1e reads/issue", base + 2, base + 3)
    row("  R1e wait", base + 3, base + 4)
    row("  M1e", base + 4, base + 5)
    row("  R2e reads/issue", base + 5, base + 6)
    row("  R2e wait", base + 6, base + 7)
    row("  M2e", base + 7, base + 8)
    row("  R1o reads/issue", base + 8, base + 9)
    row("  R1o wait", base + 9, base + 10)
    row("  M1o", base + 10, base + 11)
    row("  R2o reads/issue", base + 11, base + 12)
    row("  R2o wait", base + 12, base + 13)
    row("  M2o", base + 13, base + 14)
    row("  whole iteration", base + 2, base + 14)
print("-- inside the four MFMA phases of the plain iteration: [lgkmcnt done -> barrier passed] [barrier -> 32 MFMAs issued] [-> closing barrier is the next row's start]")
for nm, b in (("M1e", 40), ("M2e", 44), ("M1o", 48), ("M2o", 52)):
    row(f"  {nm} wait at opening barrier", b, b + 1)
    row(f"  {nm} 32 MFMAs issued", b + 1, b + 2)
if t[0, 57] > t[0, 56] > 0:
    ticks = [int(t[w_, 30] - t[w_, 18]) for w_ in range(8)]
    real = [int(t[w_, 57] - t[w_, 56]) for w_ in range(8)]
    print("-- plain iteration: s_memtime ticks / s_memrealtime ticks (100 MHz) -> clock of the s_memtime counter")
    print("  " + "  ".join(f"{a}/{b} = {a / max(b, 1) / 10:.2f} GHz" for a, b in zip(ticks, real)))
