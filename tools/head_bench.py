"""Decode-head convolution micro-benchmark: every ConvTranspose / Conv2d 3x3 pass of the head (forward, data gradient, weight
gradient) at the model's stage shapes, with the conv8 engine on and off (IG_CONV8 is read per call).

    python tools/head_bench.py --batch 216 --temporal 1 [--dim 768] [--only fwd,dgrad,wgrad] [--bn 192]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "instageo-e2e-geospatial-ml_amd"))
import torch  # noqa: E402

from instageo_amd import ops  # noqa: E402
from instageo_amd.ops import BT  # noqa: E402

dev = "cuda"


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def ab(name, fl, fn, modes):
    out = []
    for label, env in modes:
        for k, v in env.items():
            os.environ[k] = v
        t = timeit(fn)
        out.append(f"{label} {t:8.1f} us {fl / t / 1e6:6.0f} TF [{ops.last_kernel()[:44]}]")
        for k in env:
            os.environ.pop(k, None)
    print(f"{name:34s} " + " | ".join(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=216)
    ap.add_argument("--temporal", type=int, default=1)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--only", default="fwd,dgrad,wgrad")
    ap.add_argument("--split", action="store_true")
    ap.add_argument("--stages", default="0,1,2,3")
    ap.add_argument("--force", action="store_true", help="third column: both engines forced for every covered shape")
    a = ap.parse_args()
    B, sp = a.batch, a.split
    only = set(a.only.split(","))
    dims = [a.dim * a.temporal // 2**i for i in range(5)]
    modes = [("old", {"IG_CONV8": "0", "IG_WGRAD8_CONV": "0"}), ("new", {"IG_CONV8": "1", "IG_WGRAD8_CONV": "1"})]
    if a.force:
        modes.append(("forced", {"IG_CONV8": "2", "IG_WGRAD8_CONV": "2"}))

    def mk(*shape):
        return BT.from_float(torch.randn(*shape, device=dev) * 0.5, sp)

    for i in [int(s) for s in a.stages.split(",")]:
        Hs, Ci, Co = 14 * 2**i, dims[i], dims[i + 1]
        Hu = 2 * Hs
        fl = 2.0 * B * Hs * Hs * Ci * Co * 9
        x, w = mk(B, Hs, Hs, Ci), BT.from_float(torch.randn(Co, 9, Ci, device=dev) * 0.05, sp)
        bias = torch.zeros(Co, device=dev)
        u = BT.empty((B, Hu, Hu, Co), sp, dev)
        if "fwd" in only:
            ab(f"convT fwd   {Hs:3d}^2 {Ci:4d}->{Co:4d}", fl, lambda: ops.convT_fwd(x, w, bias, u, B, Hs, Hs, Ci, Co, seed=1, p=0.1), modes)
        dx = BT.empty((B, Hs, Hs, Ci), sp, dev)
        if "dgrad" in only:
            ab(f"convT dgrad {Hs:3d}^2 {Ci:4d}->{Co:4d}", fl, lambda: ops.convT_dgrad(u, w, dx, B, Hs, Hs, Ci, Co), modes)
        if "wgrad" in only:
            dw, db = torch.zeros(Co, 9, Ci, device=dev), torch.zeros(Co, device=dev)
            ab(f"convT wgrad {Hs:3d}^2 {Ci:4d}->{Co:4d}", fl, lambda: ops.convT_wgrad(u, x, dw, B, Hs, Hs, Ci, Co, dbias=db), modes)
            del dw
        del x, dx
        fl = 2.0 * B * Hu * Hu * Co * Co * 9
        w2 = BT.from_float(torch.randn(Co, 9, Co, device=dev) * 0.05, sp)
        cv = BT.empty((B, Hu, Hu, Co), sp, dev)
        if "fwd" in only:
            ab(f"conv  fwd   {Hu:3d}^2 {Co:4d}", fl, lambda: ops.conv3x3_fwd(u, w2, bias, cv, B, Hu, Hu, Co, Co), modes)
        if "dgrad" in only:
            ab(f"conv  dgrad {Hu:3d}^2 {Co:4d}", fl, lambda: ops.conv3x3_dgrad(cv, w2, u, B, Hu, Hu, Co, Co, seed=1, p=0.1), modes)
        if "wgrad" in only:
            dw = torch.zeros(Co, 9, Co, device=dev)
            ab(f"conv  wgrad {Hu:3d}^2 {Co:4d}", fl, lambda: ops.conv3x3_wgrad(cv, u, dw, B, Hu, Hu, Co, Co), modes)
            del dw
        del u, cv
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
