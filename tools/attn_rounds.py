import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "instageo-e2e-geospatial-ml_amd"))
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
for B, N, H in [(8, 197, 4), (64, 197, 4), (128, 197, 4), (192, 197, 4), (256, 197, 4), (324, 197, 4)]:
    qkv = BT(torch.randn(B, N, 3 * H * 64, device=dev).bfloat16())
    out = BT.empty((B, N, H * 64), False, dev); lse = torch.empty(B * H * N, device=dev)
    dout = BT(torch.randn(B, N, H * 64, device=dev).bfloat16()); dqkv = BT.empty((B, N, 3 * H * 64), False, dev)
    delta = torch.empty(B * H * N, device=dev)
    ops.attention_fwd(qkv, out, lse, B, N, H)
    r = []
    for f in ("1", "0"):
        os.environ["IG_ATTN2_FUSED"] = f
        r.append(timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H)))
    tf = timeit(lambda: ops.attention_fwd(qkv, out, lse, B, N, H))
    print(f"WGs {B*H:5d}: fused {r[0]:7.1f} us  two-pass {r[1]:7.1f} us  fwd {tf:7.1f}")
