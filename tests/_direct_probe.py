"""Helper of test_gpu_direct_vs_gemm.py: runs the narrow head-stage convolutions at the BASELINE image size on one fixed
seeded input and writes per-op digests (sum, sum of squares, strided samples) as JSON.  Run once with IG_CONV_DIRECT=1 and once
with IG_CONV_DIRECT=0 (the library reads the variable once per process)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instageo-e2e-geospatial-ml_amd"))
import torch  # noqa: E402

from instageo_amd import ops  # noqa: E402
from instageo_amd.ops import BT  # noqa: E402

dev = "cuda"
B, H = int(sys.argv[2]), int(sys.argv[3])


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return BT((torch.randn(*shape, generator=g) * scale).to(dev).bfloat16())


def digest(t):
    t = t.double().flatten()
    step = max(1, t.numel() // 4099)
    return {"sum": t.sum().item(), "sq": (t * t).sum().item(), "samples": t[::step][:4096].cpu().tolist()}


out = {}
# 3x3, 48 -> 48 at (H, H): forward (+bias), data gradient with the dropout mask, weight gradient
x, w = rnd(B, H, H, 48, seed=1), rnd(48, 9, 48, seed=2, scale=0.05)
bias = torch.linspace(-0.5, 0.5, 48, device=dev)
y = BT.empty((B, H, H, 48), False, dev)
ops.conv3x3_fwd(x, w, bias, y, B, H, H, 48, 48)
out["conv_fwd"] = digest(y.float())
dy = rnd(B, H, H, 48, seed=3)
dx = BT.empty((B, H, H, 48), False, dev)
ops.conv3x3_dgrad(dy, w, dx, B, H, H, 48, 48, seed=77, p=0.1)
out["conv_dgrad"] = digest(dx.float())
dw = torch.zeros(48, 9, 48, device=dev)
ops.conv3x3_wgrad(dy, x, dw, B, H, H, 48, 48)
out["conv_wgrad"] = digest(dw)
# 96 channels at (H/2, H/2): forward, data gradient (dropout), weight gradient
h2 = H // 2
x96, dy96 = rnd(B, h2, h2, 96, seed=4), rnd(B, h2, h2, 96, seed=5)
w96 = rnd(96, 9, 96, seed=7, scale=0.04)
bias96 = torch.linspace(-0.5, 0.5, 96, device=dev)
y96 = BT.empty((B, h2, h2, 96), False, dev)
ops.conv3x3_fwd(x96, w96, bias96, y96, B, h2, h2, 96, 96)
out["conv_fwd96"] = digest(y96.float())
dx96 = BT.empty((B, h2, h2, 96), False, dev)
ops.conv3x3_dgrad(dy96, w96, dx96, B, h2, h2, 96, 96, seed=79, p=0.1)
out["conv_dgrad96"] = digest(dx96.float())
dw96 = torch.zeros(96, 9, 96, device=dev)
ops.conv3x3_wgrad(dy96, x96, dw96, B, h2, h2, 96, 96)
out["conv_wgrad96"] = digest(dw96)
# ConvTranspose 96 -> 48, (H/2, H/2) -> (H, H): forward with dropout, weight gradient
wt = rnd(48, 9, 96, seed=6, scale=0.05)
yt = BT.empty((B, H, H, 48), False, dev)
ops.convT_fwd(x96, wt, bias, yt, B, h2, h2, 96, 48, seed=78, p=0.1)
out["convT_fwd"] = digest(yt.float())
dxt = BT.empty((B, h2, h2, 96), False, dev)
ops.convT_dgrad(dy, wt, dxt, B, h2, h2, 96, 48)
out["convT_dgrad"] = digest(dxt.float())
dwt = torch.zeros(48, 9, 96, device=dev)
ops.convT_wgrad(dy, x96, dwt, B, h2, h2, 96, 48)
out["convT_wgrad"] = digest(dwt)
torch.cuda.synchronize()
json.dump(out, open(sys.argv[1], "w"))
