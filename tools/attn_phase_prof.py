"""Phase timing of attn2_bwd_fused_kernel from a -DIG_A2_PROF build of the library (IG_HIP_LIB=.../libinstageo_hip_prof.so):
s_memtime at the phase boundaries of the seven waves of one workgroup.

    IG_HIP_LIB=instageo-e2e-geospatial-ml_amd/instageo_amd/libinstageo_hip_prof.so python tools/attn_phase_prof.py [B]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "instageo-e2e-geospatial-ml_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from instageo_amd import _lib, ops  # noqa: E402
from instageo_amd.ops import BT  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 216
N, H = 197, 12
dev = "cuda"
qkv = BT.from_float(torch.randn(B, N, 3 * H * 64, device=dev), False)
out = BT.empty((B, N, H * 64), False, dev)
lse = torch.empty(B, H, N, device=dev)
ops.attention_fwd(qkv, out, lse, B, N, H)
dout = BT.from_float(torch.randn(B, N, H * 64, device=dev), False)
dqkv = BT.empty((B, N, 3 * H * 64), False, dev)
delta = torch.empty(B * H * N, device=dev)
dbias = torch.zeros(3 * H * 64, device=dev)
for _ in range(3):
    ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H, dbias=dbias)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (8 * 48))()
lib.ig_debug_a2prof.argtypes = [ctypes.c_void_p]
assert lib.ig_debug_a2prof(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(8, 48).astype(np.int64)
t0 = t[:7, 0].min()
names = {0: "loads issued + delta", 1: "vmcnt(0)", 2: "barrier (images ready)", 3: "K^T frags + barrier", 46: "dQ write-out", 47: "bias gradient"}
print("wave:      " + "".join(f"{w:8d}" for w in range(7)) + "   (cycles since the first wave's first mark)")
def row(label, idx):
    print(f"{label:28s}" + "".join(f"{int(t[w, idx] - t0):8d}" for w in range(7)))
for i in (0, 1, 2, 3):
    row(names[i], i)
for s in range(7):
    for k, nm in enumerate(("step start", "S, dP MFMAs done", "softmax / dS done", "pack + dV, dK MFMAs done", "dQ MFMA + image RMW done", "barrier passed")):
        row(f"step {s}: {nm}", 4 + s * 6 + k)
row(names[46], 46)
row("bias: dO column sums", 40)
row("bias: shuffles + LDS", 41)
row("bias: barrier", 42)
row(names[47], 47)
d = np.zeros((7, 6))
for s in range(7):
    for k in range(6):
        a = t[:7, 4 + s * 6 + k] if k else t[:7, 4 + s * 6]
        prev = t[:7, 4 + s * 6 + k - 1] if k else (t[:7, 3] if s == 0 else t[:7, 9 + (s - 1) * 6])
        d[:, k] += a - prev
print("\nmean cycles per step and wave by phase: " + ", ".join(f"{nm} {d[:, k].mean() / 7:.0f}" for k, nm in enumerate(("(entry)", "S/dP MFMA", "softmax", "pack+dV/dK", "dQ+RMW", "barrier wait"))))
print(f"whole kernel of this workgroup: {int(t[:7, 47].max() - t0)} cycles; steps: {int(t[:7, 45].max() - t[:7, 3].min())}")
