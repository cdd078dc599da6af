"""What the box's HBM delivers to trivial kernels (the practical ceiling the streaming kernels are judged against):
copy (read + write), read-only sum, write-only fill on 1 GiB tensors; python tools/hbm_ceiling.py"""
import torch

x = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
y = torch.empty_like(x)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


gib = x.numel() * 4
print(f"copy  (1 GiB read + 1 GiB write): {2 * gib / timed(lambda: y.copy_(x)) / 1e12:.2f} TB/s")
print(f"sum   (1 GiB read):               {gib / timed(lambda: x.sum()) / 1e12:.2f} TB/s")
print(f"fill  (1 GiB write):              {gib / timed(lambda: y.fill_(1.0)) / 1e12:.2f} TB/s")
print(f"add   (2 GiB read + 1 GiB write): {3 * gib / timed(lambda: torch.add(x, y, out=y)) / 1e12:.2f} TB/s")
xb = x.view(torch.int32)[: x.numel() // 2].view(torch.bfloat16)
yb = torch.empty_like(xb)
print(f"copy bf16 (0.5 + 0.5 GiB):        {gib / timed(lambda: yb.copy_(xb)) / 1e12:.2f} TB/s")
