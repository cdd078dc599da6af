"""What the vendor GEMM (hipBLASLt / rocBLAS behind torch.nn.functional.linear) reaches on the encoder's shapes.

Calibration only: tells how far gemm8 is from the best hand-scheduled assembly on the same box.  Not part of the product path.
usage: python tools/vendor_gemm_ceiling.py [M]
"""
import sys
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 42552
shapes = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]
dev = torch.device("cuda:0")
for name, N, K in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    for _ in range(5):
        y = torch.nn.functional.linear(x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(3):
        e0.record()
        for _ in range(20):
            y = torch.nn.functional.linear(x, w)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{name:6s} M={M} N={N:5d} K={K:5d}: vendor linear {best:7.1f} us ({2.0 * M * N * K / best / 1e6:6.0f} TF/s)")
