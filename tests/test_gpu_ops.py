"""Per-kernel parity tests (-m gpu): every C-ABI entry point against a float64 torch restatement of the op.

Inputs are first rounded to what the kernel actually sees (bf16, or bf16 hi+lo in split mode), so the only
difference left is fp32 accumulation order: tolerances are tight.  Shapes include ragged tails.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from instageo_amd import ops  # noqa: E402
from instageo_amd.ops import BT  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + int(np.prod(shape)) % 9973)
    return (torch.randn(*shape, generator=g) * scale).float()


def bt(x, split):
    b = BT.from_float(x.to(DEV), split)
    return b, b.float().double().cpu()  # (device tensor, exact value the kernel sees)


def close(got, ref, rtol, atol=None, what=""):
    got = got.double().cpu()
    ref = ref.double()
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    tol = rtol * scale if atol is None else atol
    assert err <= tol, f"{what}: max err {err:.3e} > tol {tol:.3e} (scale {scale:.3e})"


def tol_out(split):  # output rounded to bf16 (or bf16x2)
    return 2e-5 if split else 6e-3


SPLITS = [False, True]


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (128, 128, 64), (77, 40, 8), (1000, 384, 1536)])
def test_linear_fwd(split, M, N, K):
    x, xr = bt(rnd(M, K, seed=1), split)
    w, wr = bt(rnd(N, K, seed=2, scale=K**-0.5), split)
    b = rnd(N, seed=3).to(DEV)
    y = BT.empty((M, N), split, DEV)
    pre = BT.empty((M, N), split, DEV)
    ops.linear_fwd(x, w, b, y, M, N, K, act=0)
    ref = xr @ wr.t() + b.double().cpu()
    close(y.float(), ref, tol_out(split), what="linear")
    ops.linear_fwd(x, w, b, y, M, N, K, act=1, pre=pre)  # training form: also saves gelu'(pre-activation)
    rr = ref.clone().requires_grad_(True)
    (dref,) = torch.autograd.grad(F.gelu(rr).sum(), rr)
    close(pre.float(), dref, tol_out(split), what="linear saved gelu'")
    close(y.float(), F.gelu(ref), tol_out(split), what="linear gelu")
    ops.linear_fwd(x, w, b, y, M, N, K, act=1)  # inference form
    close(y.float(), F.gelu(ref), tol_out(split), what="linear gelu (no save)")


def test_linear_big_shapes_repeatable():
    """Model-sized GEMMs (the shapes routed to the ping-pong v5 engine): repeated launches are bit-identical (race screen:
    an LDS-DMA / barrier ordering bug shows up as rare differing tiles) and agree with an fp32 matmul of the bf16 inputs."""
    M, D = 4300, 768
    # dgrad without elementwise factor: dx[M, K] = dy[M, N] @ w[N, K]
    for N, K in [(2304, 768), (768, 3072)]:
        dy, dyr = bt(rnd(M, N, seed=11), False)
        w, wr = bt(rnd(N, K, seed=12, scale=N**-0.5), False)
        dx = BT.empty((M, K), False, DEV)
        ops.linear_dgrad(dy, w, dx, M, N, K)
        first = dx.hi.clone()
        close(dx.float(), dyr @ wr, tol_out(False), what="big dgrad")
        for _ in range(15):
            dx.hi.zero_()
            ops.linear_dgrad(dy, w, dx, M, N, K)
            assert torch.equal(dx.hi, first), "dgrad differs between identical launches"
    # residual GEMM with the long reduction (fc2): out = resid + x[M, 4D] @ w[D, 4D]^T + b
    x, xr = bt(rnd(M, 4 * D, seed=13), False)
    w, wr = bt(rnd(D, 4 * D, seed=14, scale=(4 * D) ** -0.5), False)
    b = rnd(D, seed=15).to(DEV)
    res = rnd(M, D, seed=16).to(DEV)
    out = torch.empty_like(res)
    ops.linear_residual_fwd(x, w, b, res, out, M, D, 4 * D)
    first = out.clone()
    close(out, res.double().cpu() + xr @ wr.t() + b.double().cpu(), 2e-5, what="big residual")
    for _ in range(15):
        out.zero_()
        ops.linear_residual_fwd(x, w, b, res, out, M, D, 4 * D)
        assert torch.equal(out, first), "residual GEMM differs between identical launches"


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("M,N,K", [(1000, 512, 256), (256, 256, 128), (2300, 768, 768), (21276, 2304, 768), (5000, 768, 3072)])
def test_linear_v8_engine(split, M, N, K, monkeypatch):
    """The 256 x 256 x 64 8-phase engine (gemm8.hip; forced for every covered shape): qkv / fc1 (+GELU, +saved gelu') /
    proj / fc2 forms against float64, ragged last row block, repeated launches bit-identical (LDS-DMA / barrier race screen)."""
    monkeypatch.setenv("IG_GEMM8", "2")
    x, xr = bt(rnd(M, K, seed=1), split)
    w, wr = bt(rnd(N, K, seed=2, scale=K**-0.5), split)
    b = rnd(N, seed=3).to(DEV)
    y = BT.zeros((M, N), split, DEV)
    pre = BT.zeros((M, N), split, DEV)
    ref = xr @ wr.t() + b.double().cpu()
    ops.linear_fwd(x, w, b, y, M, N, K, act=0)
    close(y.float(), ref, tol_out(split), what="v8 linear")
    first = y.hi.clone()
    for _ in range(6):
        y.hi.zero_()
        ops.linear_fwd(x, w, b, y, M, N, K, act=0)
        assert torch.equal(y.hi, first), "v8 linear differs between identical launches"
    monkeypatch.setenv("IG_GEMM8", "0")
    y0 = BT.zeros((M, N), split, DEV)
    ops.linear_fwd(x, w, b, y0, M, N, K, act=0)  # the generic engines on the same inputs
    close(y.float(), y0.float().double().cpu(), tol_out(split), what="v8 vs generic engine")
    monkeypatch.setenv("IG_GEMM8", "2")
    ops.linear_fwd(x, w, None, y, M, N, K, act=0)
    close(y.float(), xr @ wr.t(), tol_out(split), what="v8 linear, no bias")
    ops.linear_fwd(x, w, b, y, M, N, K, act=1, pre=pre)
    rr = ref.clone().requires_grad_(True)
    (dref,) = torch.autograd.grad(F.gelu(rr).sum(), rr)
    close(y.float(), F.gelu(ref), tol_out(split), what="v8 gelu")
    close(pre.float(), dref, tol_out(split), what="v8 saved gelu'")
    ops.linear_fwd(x, w, b, y, M, N, K, act=1)
    close(y.float(), F.gelu(ref), tol_out(split), what="v8 gelu (no save)")
    res = rnd(M, N, seed=4).to(DEV)
    out = torch.zeros_like(res)
    ops.linear_residual_fwd(x, w, b, res, out, M, N, K)
    rref = res.double().cpu() + ref
    close(out, rref, 2e-5 if split else 1e-5, what="v8 residual")
    ops.linear_residual_fwd(x, w, b, res, res, M, N, K)  # in place, as the engine uses it
    close(res, rref, 2e-5 if split else 1e-5, what="v8 residual in-place")


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("M,N,K", [(1000, 512, 256), (2300, 768, 768), (21276, 2304, 768), (5000, 768, 3072), (700, 1024, 1024), (600, 1024, 4096),
                                   (900, 1280, 1280), (257, 256, 384)])
def test_linear_v4_engine(split, M, N, K, monkeypatch):
    """The 4-wave / one-wave-per-SIMD engine with the generated K-loop (gemm4.hip + gen_gemm4.py), forced for every covered shape
    (IG_GEMM4=2), plain bf16 operands and the split precision mode on paired K-tiles: qkv / fc1 (+GELU, +saved gelu') / proj / fc2 forms and
    the data gradient with the gelu' factor + fused column sums, against float64 on the same rounded operands; K = 256 (no middle loop trip)
    .. 4096, the 300M / 600M widths (K = 1024 / 4096 / 1280), a ragged last row block and a one-row last block; repeated launches
    bit-identical (LDS-DMA / barrier / AGPR read-out race screen); equal to the 8-phase engine's result to the output rounding."""
    monkeypatch.setenv("IG_GEMM8", "2")
    monkeypatch.setenv("IG_GEMM4", "2")
    tag = ",2>" if split else ">"  # the paired instances carry NSEG = 2 as a fourth template field
    x, xr = bt(rnd(M, K, seed=1), split)
    w, wr = bt(rnd(N, K, seed=2, scale=K**-0.5), split)
    b = rnd(N, seed=3).to(DEV)
    y = BT.zeros((M, N), split, DEV)
    pre = BT.zeros((M, N), split, DEV)
    ref = xr @ wr.t() + b.double().cpu()
    ops.linear_fwd(x, w, b, y, M, N, K, act=0)
    assert ops.last_kernel().startswith("gemm4_kernel<0,0,false") and ops.last_kernel().endswith(tag), ops.last_kernel()
    close(y.float(), ref, tol_out(split), what="v4 linear")
    first = (y.hi.clone(), None if not split else y.lo.clone())
    for _ in range(6):
        y.hi.zero_()
        ops.linear_fwd(x, w, b, y, M, N, K, act=0)
        assert torch.equal(y.hi, first[0]) and (not split or torch.equal(y.lo, first[1])), "v4 linear differs between identical launches"
    monkeypatch.setenv("IG_GEMM4", "0")
    y8 = BT.zeros((M, N), split, DEV)
    ops.linear_fwd(x, w, b, y8, M, N, K, act=0)
    assert ops.last_kernel().startswith("gemm8_kernel"), ops.last_kernel()
    close(y.float(), y8.float().double().cpu(), tol_out(split), what="v4 vs v8")
    monkeypatch.setenv("IG_GEMM4", "2")
    ops.linear_fwd(x, w, None, y, M, N, K, act=0)
    close(y.float(), xr @ wr.t(), tol_out(split), what="v4 linear, no bias")
    ops.linear_fwd(x, w, b, y, M, N, K, act=1, pre=pre)
    # the split mode saves gelu' in two halves (four staging slabs): that kind stays on the 8-phase engine
    assert ops.last_kernel().startswith("gemm8_kernel" if split else "gemm4_kernel<0,1,true"), ops.last_kernel()
    rr = ref.clone().requires_grad_(True)
    (dref,) = torch.autograd.grad(F.gelu(rr).sum(), rr)
    close(y.float(), F.gelu(ref), tol_out(split), what="v4 gelu")
    close(pre.float(), dref, tol_out(split), what="v4 saved gelu'")
    ops.linear_fwd(x, w, b, y, M, N, K, act=1)
    assert ops.last_kernel().startswith("gemm4_kernel<0,1,false") and ops.last_kernel().endswith(tag), ops.last_kernel()
    close(y.float(), F.gelu(ref), tol_out(split), what="v4 gelu (no save)")
    res = rnd(M, N, seed=4).to(DEV)
    out = torch.zeros_like(res)
    ops.linear_residual_fwd(x, w, b, res, out, M, N, K)
    assert ops.last_kernel().startswith("gemm4_kernel<1") and ops.last_kernel().endswith(tag), ops.last_kernel()
    rref = res.double().cpu() + ref
    close(out, rref, 2e-5 if split else 1e-5, what="v4 residual")
    first = out.clone()
    for _ in range(4):
        out.zero_()
        ops.linear_residual_fwd(x, w, b, res, out, M, N, K)
        assert torch.equal(out, first), "v4 residual differs between identical launches"
    ops.linear_residual_fwd(x, w, b, res, res, M, N, K)  # in place, as the engine uses it
    close(res, rref, 2e-5 if split else 1e-5, what="v4 residual in-place")
    # data gradient through the transposed weight copy: dx = (dy @ w) * dact, fused column sums
    dy, dyr = bt(rnd(M, N, seed=5), split)
    wt = BT.empty((K, N), split, DEV)          # (K, N): the K-contiguous operand of the dgrad GEMM  (here: dx is (M, K))
    wt.hi.copy_(w.hi.t())
    if split:
        wt.lo.copy_(w.lo.t())
    dx = BT.zeros((M, K), split, DEV)
    fac, facr = bt(rnd(M, K, seed=8), split)
    cs = torch.zeros(K, device=DEV)
    if K % 256 == 0 and N >= 256:
        ops.linear_dgrad(dy, None, dx, M, N, K, pre=fac, colsum=cs, wt=wt)
        assert ops.last_kernel().startswith("gemm4_kernel<2") and ops.last_kernel().endswith(tag), ops.last_kernel()
        want = (dyr @ wr) * facr
        close(dx.float(), want, tol_out(split), what="v4 dgrad*dact")
        close(cs, want.sum(0), 3e-5, what="v4 fused column sums")
        ops.linear_dgrad(dy, None, dx, M, N, K, wt=wt)
        assert ops.last_kernel().startswith("gemm4_kernel<0,0") and ops.last_kernel().endswith(tag), ops.last_kernel()
        close(dx.float(), dyr @ wr, tol_out(split), what="v4 plain dgrad")


@pytest.mark.parametrize("split", SPLITS)
def test_linear_residual(split):
    M, N, K = 333, 256, 1024
    x, xr = bt(rnd(M, K, seed=1), split)
    w, wr = bt(rnd(N, K, seed=2, scale=K**-0.5), split)
    b = rnd(N, seed=3).to(DEV)
    res = rnd(M, N, seed=4).to(DEV)
    out = torch.empty_like(res)
    ops.linear_residual_fwd(x, w, b, res, out, M, N, K)
    ref = res.double().cpu() + xr @ wr.t() + b.double().cpu()
    close(out, ref, 2e-5 if split else 1e-5, what="residual")
    # in-place (out aliases resid) as the engine uses it
    ops.linear_residual_fwd(x, w, b, res, res, M, N, K)
    close(res, ref, 2e-5 if split else 1e-5, what="residual in-place")


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (197, 768, 256), (1000, 64, 72), (4100, 2304, 768), (2100, 1280, 1536), (21168, 768, 3072)])
def test_linear_dgrad_wgrad(split, M, N, K):
    dy, dyr = bt(rnd(M, N, seed=5), split)
    w, wr = bt(rnd(N, K, seed=6, scale=N**-0.5), split)
    x, xr = bt(rnd(M, K, seed=7), split)
    dx = BT.empty((M, K), split, DEV)
    ops.linear_dgrad(dy, w, dx, M, N, K)
    close(dx.float(), dyr @ wr, tol_out(split), what="dgrad")
    pre, gref = bt(rnd(M, K, seed=8), split)  # the elementwise factor the forward saved (gelu' there; any tensor here)
    cs = torch.zeros(K, device=DEV)
    ops.linear_dgrad(dy, w, dx, M, N, K, pre=pre, colsum=cs)
    close(dx.float(), (dyr @ wr) * gref, tol_out(split), what="dgrad*dact")
    close(cs, ((dyr @ wr) * gref).sum(0), 3e-5, what="fused column sums (bias grad)")
    dw = torch.zeros(N, K, device=DEV)
    ops.linear_wgrad(dy, x, dw, M, N, K)
    close(dw, dyr.t() @ xr, 3e-5 if split else 2e-5, what="wgrad")
    ops.linear_wgrad(dy, x, dw, M, N, K)  # accumulates
    close(dw, 2 * (dyr.t() @ xr), 3e-5 if split else 2e-5, what="wgrad accumulate")


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("M,N,K,force", [(1000, 256, 512, "2"), (21276, 768, 3072, "1"), (5000, 2304, 768, "2"), (333, 192, 64, "1"), (2300, 768, 768, "2")])
def test_linear_dgrad_wt(split, M, N, K, force, monkeypatch):
    """dx = dy @ w through the TRANSPOSED weight copy (ig_transpose_bf16 + ig_linear_dgrad_wt): the K-contiguous form on the
    256 x 256 x 64 engine (forced for every covered shape with IG_GEMM8=2) or the generic engines, plain and with the saved
    gelu' factor + fused column sums; bit-identical across launches."""
    monkeypatch.setenv("IG_GEMM8", force)
    dy, dyr = bt(rnd(M, N, seed=5), split)
    w, wr = bt(rnd(N, K, seed=6, scale=N**-0.5), split)
    wt = BT.empty((K, N), split, DEV)
    if N % 64 == 0 and K % 64 == 0:
        ops.transpose_bf16(w, wt, N, K)
    else:
        wt = BT(w.hi.t().contiguous(), None if w.lo is None else w.lo.t().contiguous())
    assert torch.equal(wt.hi, w.hi.t().contiguous()) and (not split or torch.equal(wt.lo, w.lo.t().contiguous())), "transpose"
    dx = BT.zeros((M, K), split, DEV)
    ops.linear_dgrad(dy, None, dx, M, N, K, wt=wt)
    close(dx.float(), dyr @ wr, tol_out(split), what="dgrad (wt)")
    first = dx.hi.clone()
    for _ in range(4):
        dx.hi.zero_()
        ops.linear_dgrad(dy, None, dx, M, N, K, wt=wt)
        assert torch.equal(dx.hi, first), "dgrad (wt) differs between identical launches"
    pre, gref = bt(rnd(M, K, seed=8), split)
    cs = torch.zeros(K, device=DEV)
    ops.linear_dgrad(dy, None, dx, M, N, K, pre=pre, colsum=cs, wt=wt)
    close(dx.float(), (dyr @ wr) * gref, tol_out(split), what="dgrad*dact (wt)")
    close(cs, ((dyr @ wr) * gref).sum(0), 3e-5, what="fused column sums (wt)")
    ops.linear_dgrad(dy, None, dx, M, N, K, pre=pre, wt=wt)  # without the column sums
    close(dx.float(), (dyr @ wr) * gref, tol_out(split), what="dgrad*dact, no colsum (wt)")


def test_reserved_cus_and_the_persistent_grids(monkeypatch):
    """world > 1: distributed.attach_data_parallel asks for 8 CUs for RCCL's kernels.  The tile-walking GEMMs launch the fewest
    workgroups that keep the number of rounds (252 tiles -> 252 workgroups: 4 CUs stay free) and honour the reservation only
    when it does not add a round (here it would double the launch); IG_RESERVED_STRICT=1 forces it.  Checked on the launched
    grid; the result does not depend on it."""
    from instageo_amd import _lib

    lib = _lib.load()
    M, N, K = 21276, 768, 768
    x, xr = bt(rnd(M, K, seed=1), False)
    w, wr = bt(rnd(N, K, seed=2, scale=K**-0.5), False)
    y = BT.zeros((M, N), False, DEV)
    ref = None
    try:
        for reserve, strict, engine in ((0, "0", "1"), (8, "0", "1"), (8, "1", "1"), (8, "0", "0"), (8, "1", "0"), (0, "0", "0")):
            monkeypatch.setenv("IG_GEMM8", engine)
            monkeypatch.setenv("IG_RESERVED_STRICT", strict)
            ops.set_reserved_cus(reserve)
            assert ops.reserved_cus() == reserve
            ops.linear_fwd(x, w, None, y, M, N, K)
            per_cu = 1 if engine == "1" else 2  # gemm8: one workgroup per CU (84 x 3 tiles); gemm2: two (84 x 6 tiles)
            grid = lib.ig_last_grid()
            if reserve and strict == "1":
                assert 0 < grid <= per_cu * (256 - reserve), (reserve, strict, engine, grid)
            else:
                assert grid == per_cu * 252, (reserve, strict, engine, grid)
            if ref is None:
                ref = y.hi.clone()
                close(y.float(), xr @ wr.t(), tol_out(False), what="reserved-CU gemm")
            elif engine == "1":
                assert torch.equal(y.hi, ref)
    finally:
        ops.set_reserved_cus(0)


def test_transpose_bf16_batched():
    R, C, nb = 192, 128, 3
    src = BT.from_float(torch.randn(nb * R * C + 64, device=DEV), True)  # matrices 8 elements apart from contiguous
    dst = BT.zeros((nb * R * C + 128,), True, DEV)
    ops.transpose_bf16(src, dst, R, C, nb, R * C + 8, R * C + 16)
    for b in range(nb):
        for a, o in ((src.hi, dst.hi), (src.lo, dst.lo)):
            m = a[b * (R * C + 8) : b * (R * C + 8) + R * C].view(R, C)
            t = o[b * (R * C + 16) : b * (R * C + 16) + R * C].view(C, R)
            assert torch.equal(t, m.t()), f"matrix {b}"
    with pytest.raises(Exception):
        ops.transpose_bf16(src, dst, 100, 64)


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("M,N,K", [(21276, 2304, 768), (21276, 768, 768), (4100, 768, 3072), (3152, 768, 768)])
def test_linear_wgrad_is_deterministic(split, M, N, K):
    """The reference trains with deterministic=True: split-K weight gradients go through workspace slabs + one ordered reduce
    (no float atomics), so repeated launches are bit-identical."""
    dy, dyr = bt(rnd(M, N, seed=5), split)
    x, xr = bt(rnd(M, K, seed=7), split)
    dw = torch.zeros(N, K, device=DEV)
    ops.linear_wgrad(dy, x, dw, M, N, K)
    close(dw, dyr.t() @ xr, 3e-5 if split else 2e-5, what="wgrad")
    first = dw.clone()
    for _ in range(4):
        dw.zero_()
        ops.linear_wgrad(dy, x, dw, M, N, K)
        assert torch.equal(dw, first), "weight gradient differs between identical launches"


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("M,shapes", [
    (21276, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]),  # one Block of Prithvi-100M at the benchmark batch (108 tiles)
    (3152, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]),   # the YAML's batch 16
    (197, [(256, 256), (768, 256)]),                               # fewer tokens than one K-tile pair per split
    (130, [(256, 512)]),                                           # ragged: 2 valid tokens in the second K-tile pair
    (1, [(256, 256)]),
    (6304, [(1024, 4096), (4096, 1024), (1024, 1024), (3072, 1024)]),  # Prithvi-300M Block: 192 tiles, no token split
    (1000, [(2048, 4096), (4096, 2048), (2048, 2048), (6144, 2048)]),  # more tiles (768) than CUs: several tiles per workgroup
])
def test_linear_wgrad_group(split, M, shapes):
    """Grouped weight gradients (ig_linear_wgrad_group -> gemm8w.hip) against float64 dy^T @ x of the values the kernel sees;
    accumulation into dw; bit-identical repeats (ordered split-K fold: the reference trains with deterministic=True)."""
    items, refs = [], []
    for gi, (N, K) in enumerate(shapes):
        dy, dyr = bt(rnd(M, N, seed=11 + gi), split)
        x, xr = bt(rnd(M, K, seed=31 + gi), split)
        dw = torch.full((N, K), 0.5, device=DEV)
        items.append((dy, x, dw, N, K))
        refs.append(dyr.t() @ xr)
    ops.linear_wgrad_group(items, M)
    for (dy, x, dw, N, K), ref in zip(items, refs):
        close(dw - 0.5, ref, 3e-5 if split else 2e-5, atol=None, what=f"grouped wgrad {N}x{K}")
    first = [it[2].clone() for it in items]
    for _ in range(3):
        for it in items:
            it[2].fill_(0.5)
        ops.linear_wgrad_group(items, M)
        for it, f in zip(items, first):
            assert torch.equal(it[2], f), "grouped weight gradient differs between identical launches"
    # overwrite: dW = dy^T x whatever dW held (NaN here), bit-identical to the accumulate form on a zeroed dW
    for it in items:
        it[2].zero_()
    ops.linear_wgrad_group(items, M)
    acc = [it[2].clone() for it in items]
    for it in items:
        it[2].fill_(float("nan"))
    ops.linear_wgrad_group(items, M, overwrite=True)
    for it, a in zip(items, acc):
        assert torch.equal(it[2], a), "overwrite form differs from accumulate-into-zero"


@pytest.mark.parametrize("M,shapes", [
    (42336, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]),  # the benchmark batch (B = 216): 2 token splits + aligned remainders
    (21276, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]),
    (42336 + 37, [(1280, 1280), (256, 5120)]),                     # ragged last K-tile, other pitches
    (6304, [(1024, 4096), (4096, 1024), (1024, 1024), (3072, 1024)]),
])
@pytest.mark.parametrize("split", SPLITS)
def test_linear_wgrad_group_four_wave_kernel(split, M, shapes, monkeypatch):
    """gemm4w_kernel (4 waves, generated-assembly K-loop, descriptor-bounded ragged tails; plain and paired-split forms) takes the grouped
    launches of the linears and equals the 8-wave kernel BIT FOR BIT: same segments, same MFMA instruction in the same token (and hi / lo
    product) order, same ordered fold."""
    items, refs = [], []
    for gi, (N, K) in enumerate(shapes):
        dy, dyr = bt(rnd(M, N, seed=51 + gi), split)
        x, xr = bt(rnd(M, K, seed=71 + gi), split)
        items.append((dy, x, torch.zeros(N, K, device=DEV), N, K))
        refs.append(dyr.t() @ xr)
    out = {}
    for arm in ("0", "1"):
        monkeypatch.setenv("IG_GEMM4", arm)
        for it in items:
            it[2].fill_(float("nan"))
        ops.linear_wgrad_group(items, M, overwrite=True)
        want = f"gemm4w_kernel<{'true' if split else 'false'}>" if arm == "1" else f"gemm8w_kernel<{2 if split else 1},0,4,2>"
        assert ops.last_kernel() == want, ops.last_kernel()
        out[arm] = [it[2].clone() for it in items]
    for a, b, ref, (N, K) in zip(out["0"], out["1"], refs, shapes):
        assert torch.equal(a, b), f"gemm4w differs from gemm8w on dW {N}x{K}"
        close(b, ref, 3e-5 if split else 2e-5, atol=None, what=f"gemm4w {N}x{K}")


def test_linear_wgrad_group_fallback_shapes():
    """Shapes the 8-phase engine does not cover (N or K not a multiple of 256) run one ig_linear_wgrad per GEMM."""
    M = 500
    items, refs = [], []
    for gi, (N, K) in enumerate([(192, 64), (256, 256), (40, 72)]):
        dy, dyr = bt(rnd(M, N, seed=3 + gi), False)
        x, xr = bt(rnd(M, K, seed=9 + gi), False)
        items.append((dy, x, torch.zeros(N, K, device=DEV), N, K))
        refs.append(dyr.t() @ xr)
    ops.linear_wgrad_group(items, M)
    for it, ref in zip(items, refs):
        close(it[2], ref, 2e-5, what="grouped wgrad fallback")
    for it in items:  # overwrite on the fallback path: the library clears dW first
        it[2].fill_(float("nan"))
    ops.linear_wgrad_group(items, M, overwrite=True)
    for it, ref in zip(items, refs):
        close(it[2], ref, 2e-5, what="grouped wgrad fallback, overwrite")


def test_zero_ranges_table():
    """ig_zero_ranges: base[lo:hi] = 0 for a device table of ranges (odd starts and lengths), nothing else touched."""
    base = torch.arange(1, 20001, dtype=torch.float32, device=DEV)
    ranges = [(0, 5), (7, 8), (13, 4100), (4101, 4104), (9999, 20000)]
    ops.ZeroRanges(ranges, DEV).launch(base)
    ref = torch.arange(1, 20001, dtype=torch.float32)
    for a, b in ranges:
        ref[a:b] = 0
    assert torch.equal(base.cpu(), ref)
    ops.ZeroRanges([], DEV).launch(base)  # empty table: no launch


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("T", [1, 3])
def test_patch_embed(split, T):
    B, C, H, W, p, D = 2, 6, 64, 48, 16, 64
    gh, gw = H // p, W // p
    tpc = T * gh * gw
    img = rnd(B, C, T, H, W, seed=9)
    wt = rnd(D, C, 1, p, p, seed=10, scale=0.05)
    bias = rnd(D, seed=11)
    pos = rnd(1 + tpc, D, seed=12)
    patches = BT.empty((B * tpc, C * p * p), split, DEV)
    ops.patchify(img.to(DEV), p, patches)
    # exact reference of the gather (token order t,row,col ; k = c,iy,ix)
    xp = img.reshape(B, C, T, gh, p, gw, p).permute(0, 2, 3, 5, 1, 4, 6).reshape(B * tpc, C * p * p)
    close(patches.float(), xp, 1e-5 if split else 4e-3, what="patchify")
    w_bt, wr = bt(wt.reshape(D, -1), split)
    x = torch.zeros(B, 1 + tpc, D, device=DEV)
    ops.patch_embed_fwd(patches, w_bt, bias.to(DEV), pos.to(DEV), x, B, tpc, D, C * p * p)
    cls = rnd(D, seed=13)
    ops.cls_rows(x, cls.to(DEV), pos.to(DEV), B, 1 + tpc, D)
    ref = (patches.float().double().cpu() @ wr.t() + bias.double()).reshape(B, tpc, D) + pos[1:].double()
    close(x[:, 1:], ref, 2e-5, what="patch embed")
    close(x[:, 0], (cls + pos[0]).double().expand(B, D), 1e-6, what="cls rows")


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("D", [256, 768, 1024])
def test_layernorm_fwd_bwd(split, D):
    M = 197 * 2 + 3
    x = rnd(M, D, seed=14) * 2 + 0.5
    g = 1 + 0.1 * rnd(D, seed=15)
    b = 0.1 * rnd(D, seed=16)
    out = BT.empty((M, D), split, DEV)
    mean = torch.empty(M, device=DEV)
    rstd = torch.empty(M, device=DEV)
    ops.layernorm_fwd(x.to(DEV), g.to(DEV), b.to(DEV), out, mean, rstd, M, D)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.layer_norm(xd, (D,), gd, bd, 1e-5)
    close(out.float(), ref.detach(), tol_out(split), what="ln fwd")
    close(mean, x.double().mean(1), 1e-5, what="ln mean")
    dy, dyr = bt(rnd(M, D, seed=17), split)
    gx, gg, gb = torch.autograd.grad((ref * dyr).sum(), [xd, gd, bd])
    dx0 = rnd(M, D, seed=18)
    dx = dx0.clone().to(DEV)
    dxb = BT.empty((M, D), split, DEV)
    dgam = torch.zeros(D, device=DEV)
    dbet = torch.zeros(D, device=DEV)
    dcol = torch.zeros(D, device=DEV)
    ops.layernorm_bwd(dy, x.to(DEV), mean, rstd, g.to(DEV), dx, True, dxb, dgam, dbet, dcol, M, D)
    close(dx, dx0.double() + gx, 2e-5, what="ln dx")
    close(dxb.float(), dx0.double() + gx, tol_out(split), what="ln dx bf16")
    close(dgam, gg, 2e-5, what="ln dgamma")
    close(dbet, gb, 2e-5, what="ln dbeta")
    close(dcol, (dx0.double() + gx).sum(0), 5e-5, what="ln dcol")
    dx2 = torch.full((M, D), 7.0, device=DEV)
    ops.layernorm_bwd(dy, x.to(DEV), mean, rstd, g.to(DEV), dx2, False, None, None, None, None, M, D)
    close(dx2, gx, 2e-5, what="ln dx (no accumulate)")


@pytest.mark.parametrize("M", [21276, 42552, 42555])
def test_layernorm_bwd_benchmark_token_counts(M):
    """LayerNorm backward at the token counts of the benchmark batches (108 / 216 chips x 197 tokens, and a ragged count): the
    rows-per-workgroup rule changes with M (48 -> 96 rows), the result must not -- against float64 autograd on the device."""
    D = 768
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(M, D, device=DEV, generator=gen) * 2 + 0.5
    g = 1 + 0.1 * torch.randn(D, device=DEV, generator=gen)
    b = 0.1 * torch.randn(D, device=DEV, generator=gen)
    out = BT.empty((M, D), False, DEV)
    mean = torch.empty(M, device=DEV)
    rstd = torch.empty(M, device=DEV)
    ops.layernorm_fwd(x, g, b, out, mean, rstd, M, D)
    dy = BT.from_float(torch.randn(M, D, device=DEV, generator=gen), False)
    xd, gd, bd = x.double().requires_grad_(True), g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.layer_norm(xd, (D,), gd, bd, 1e-5)
    gx, gg, gb = torch.autograd.grad((ref * dy.float().double()).sum(), [xd, gd, bd])
    dx0 = torch.randn(M, D, device=DEV, generator=gen)
    dx = dx0.clone()
    dgam, dbet, dcol = (torch.zeros(D, device=DEV) for _ in range(3))
    ops.layernorm_bwd(dy, x, mean, rstd, g, dx, True, None, dgam, dbet, dcol, M, D)
    want = dx0.double() + gx
    assert (dx.double() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    for got, w, what in ((dgam, gg, "dgamma"), (dbet, gb, "dbeta"), (dcol, want.sum(0), "dcol")):
        err = (got.double() - w).abs().max().item() / max(1.0, w.abs().max().item())
        assert err <= 5e-5, (what, err)


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("T", [1, 3])
def test_layernorm_feature_layout(split, T):
    B, G, D = 2, 9, 64
    ntok = 1 + T * G
    M = B * ntok
    x = rnd(M, D, seed=19)
    g = 1 + 0.1 * rnd(D, seed=20)
    b = 0.1 * rnd(D, seed=21)
    out = BT.zeros((B, G, D * T), split, DEV)
    mean = torch.empty(M, device=DEV)
    rstd = torch.empty(M, device=DEV)
    ops.layernorm_fwd(x.to(DEV), g.to(DEV), b.to(DEV), out, mean, rstd, M, D, feat_T=T, feat_G=G, ntok=ntok)
    xd = x.double().requires_grad_(True)
    ln = F.layer_norm(xd, (D,), g.double(), b.double(), 1e-5).reshape(B, ntok, D)
    # model.py:406-413: drop cls, permute(0,2,1), reshape(B,-1,side,side) -> channel = d*T + t ; here NHWC [B][G][D*T]
    feat = ln[:, 1:, :].permute(0, 2, 1).reshape(B, D * T, G).permute(0, 2, 1)
    close(out.float(), feat.detach(), tol_out(split), what="feature layout")
    dy, dyr = bt(rnd(B, G, D * T, seed=22), split)
    (gx,) = torch.autograd.grad((feat * dyr).sum(), xd)
    dx = torch.zeros(M, D, device=DEV)
    ops.layernorm_bwd(dy, x.to(DEV), mean, rstd, g.to(DEV), dx, False, None, None, None, None, M, D, feat_T=T, feat_G=G, ntok=ntok)
    close(dx, gx, 2e-5, what="feature layout bwd")


def attn_ref(qkv, B, N, H, hd=64):
    q, k, v = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    att = ((q * hd**-0.5) @ k.transpose(-2, -1)).softmax(-1)
    return (att @ v).transpose(1, 2).reshape(B, N, H * hd)


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("N", [197, 589, 16, 33, 1, 224, 225])
def test_attention_fwd_bwd(split, N):
    B, H = 2, 3
    qkv, qr = bt(rnd(B, N, 3 * H * 64, seed=23), split)
    out = BT.empty((B, N, H * 64), split, DEV)
    lse = torch.empty(B, H, N, device=DEV)
    ops.attention_fwd(qkv, out, lse, B, N, H)
    qd = qr.clone().requires_grad_(True)
    ref = attn_ref(qd, B, N, H)
    close(out.float(), ref.detach(), 3e-5 if split else 1e-2, what="attn fwd")
    # spiked row exercises the online-softmax rescale branch
    dout, dor = bt(rnd(B, N, H * 64, seed=24), split)
    (gref,) = torch.autograd.grad((ref * dor).sum(), qd)
    # the kernel uses the O it produced (rounded); feed that back for delta
    dqkv = BT.empty((B, N, 3 * H * 64), split, DEV)
    delta = torch.empty(B * H * N, device=DEV)
    ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H)
    close(dqkv.float(), gref, 1e-4 if split else 2e-2, what="attn bwd")
    # bias gradient of the fused qkv Linear = column sums of dqkv over the tokens: fused into the single-pass backward (bf16, N <= 224:
    # Q part from the dQ image, V part = column sums of dO, K part identically zero -- the rows of dS sum to zero), one column-sum
    # pass behind the other kernels.  It ACCUMULATES into the buffer.
    dbias = torch.full((3 * H * 64,), 0.5, device=DEV)
    dq2 = BT.empty((B, N, 3 * H * 64), split, DEV)
    ops.attention_bwd(qkv, out, dout, lse, delta, dq2, B, N, H, dbias=dbias)
    assert torch.equal(dq2.hi, dqkv.hi)
    bref = gref.reshape(B * N, 3 * H * 64).sum(0)
    close(dbias - 0.5, bref, 2e-5 if split else 6e-3, what="qkv bias gradient")
    kpart = (dbias - 0.5)[H * 64 : 2 * H * 64].abs().max().item()
    assert kpart <= (1e-4 if split else 2e-2) * bref.abs().max().item(), f"K third of the qkv bias gradient should vanish, got {kpart}"


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("N,hd", [(257, 80), (769, 80), (1, 80), (16, 80), (33, 80), (197, 64)])
def test_attention_generic_head_dim(split, N, hd, monkeypatch):
    """Head dimensions other than 64 (attention_g.hip): the Prithvi-EO-2.0 600M variants run 16 heads of 80 over 257 (T = 1) or
    769 (T = 3) tokens (model.py:154-167).  hd = 64 is forced onto the same kernels (IG_ATTN_GENERIC=1) as a cross-check against
    the case the tuned kernels are tested on."""
    if hd == 64:
        monkeypatch.setenv("IG_ATTN_GENERIC", "1")
    B, H = 2, 3
    x = rnd(B, N, 3 * H * hd, seed=23)
    if N > 70:
        x[0, 70, H * hd : H * hd + hd] *= 10.0  # a spiking key in a late tile: the running maximum moves
    qkv, qr = bt(x, split)
    out = BT.empty((B, N, H * hd), split, DEV)
    lse = torch.empty(B, H, N, device=DEV)
    ops.attention_fwd(qkv, out, lse, B, N, H, hd=hd)
    # plain bf16: the LDS-staged kernels (whole head in two LDS images, three chunks at N = 769); split: the register / L2 kernels
    assert ("lds" in ops.last_kernel()) == (not split), ops.last_kernel()
    qd = qr.clone().requires_grad_(True)
    ref = attn_ref(qd, B, N, H, hd)
    close(out.float(), ref.detach(), 3e-5 if split else 1e-2, what="generic attn fwd")
    q, k, _ = qr.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    lse_ref = torch.logsumexp((q * hd**-0.5) @ k.transpose(-2, -1), -1)
    close(lse, lse_ref, 1e-5 if split else 2e-3, what="generic attn lse")
    dout, dor = bt(rnd(B, N, H * hd, seed=24), split)
    (gref,) = torch.autograd.grad((ref * dor).sum(), qd)
    dqkv = BT.empty((B, N, 3 * H * hd), split, DEV)
    delta = torch.empty(B * H * N, device=DEV)
    ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, N, H, hd=hd)
    close(dqkv.float(), gref, 1e-4 if split else 2e-2, what="generic attn bwd")


@pytest.mark.parametrize("split", SPLITS)
def test_attention_rescale_branch(split):
    B, H, N = 1, 1, 100
    x = rnd(B, N, 3 * 64, seed=25)
    x[0, 70, 64:128] *= 12.0  # key 70 spikes in a late tile for many queries
    qkv, qr = bt(x, split)
    out = BT.empty((B, N, 64), split, DEV)
    lse = torch.empty(B, H, N, device=DEV)
    ops.attention_fwd(qkv, out, lse, B, N, H)
    close(out.float(), attn_ref(qr, B, N, H), 3e-5 if split else 1e-2, what="attn rescale")


def nhwc(x):  # NCHW -> NHWC
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 14, 48, 24), (1, 28, 144, 72), (3, 7, 8, 136), (2, 21, 48, 48), (1, 48, 48, 48), (3, 5, 48, 48),
                                          (2, 21, 96, 96), (1, 18, 48, 96), (1, 35, 96, 48), (2, 13, 192, 192), (1, 9, 192, 48),
                                          (1, 1, 48, 48), (1, 16, 48, 48), (1, 3, 96, 96), (5, 2, 192, 96)])
def test_conv3x3(split, B, H, Cin, Cout):  # unsplit 48 -> 48 (fwd, dgrad) and Cin 48 / 96 (wgrad) run the halo-tile direct kernels
    W = H + 2
    x, xr = bt(nhwc(rnd(B, Cin, H, W, seed=26)), split)
    wt = rnd(Cout, Cin, 3, 3, seed=27, scale=(9 * Cin) ** -0.5)
    w, wr = bt(wt.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous(), split)
    wr_t = wr.reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    bias = rnd(Cout, seed=28)
    y = BT.empty((B, H, W, Cout), split, DEV)
    ops.conv3x3_fwd(x, w, bias.to(DEV), y, B, H, W, Cin, Cout)
    xin = xr.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wv = wr_t.clone().requires_grad_(True)
    ref = F.conv2d(xin, wv, bias.double(), padding=1)
    close(y.float(), nhwc(ref.detach()), tol_out(split), what="conv fwd")
    dy, dyr = bt(nhwc(rnd(B, Cout, H, W, seed=29)), split)
    gx, gw = torch.autograd.grad((ref * dyr.permute(0, 3, 1, 2)).sum(), [xin, wv])
    dx = BT.empty((B, H, W, Cin), split, DEV)
    ops.conv3x3_dgrad(dy, w, dx, B, H, W, Cin, Cout)
    close(dx.float(), nhwc(gx), tol_out(split), what="conv dgrad")
    dw, db = torch.zeros(Cout, 9, Cin, device=DEV), torch.full((Cout,), 0.25, device=DEV)
    ops.conv3x3_wgrad(dy, x, dw, B, H, W, Cin, Cout, dbias=db)
    close(dw, gw.permute(0, 2, 3, 1).reshape(Cout, 9, Cin), 3e-5, what="conv wgrad")
    close(db - 0.25, dyr.sum((0, 1, 2)), 3e-5, what="conv bias gradient (fused into the direct kernels / column-sum fallback)")


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("B,H,Cin,Cout,ks", [(2, 16, 64, 32, 5), (1, 30, 80, 80, 7), (2, 9, 160, 160, 5), (1, 5, 16, 8, 7), (3, 7, 24, 40, 5),
                                             (1, 32, 640, 640, 5)])
def test_conv_kxk_padding1(split, B, H, Cin, Cout, ks):
    """nn.Conv2d(kernel_size=5 | 7, padding=1) of the 600M variants' decode head (model.py:169-177): the map shrinks to H + 3 - ks;
    forward (bias), data gradient and weight / bias gradients against float64 F.conv2d."""
    W = H + 2
    Ho, Wo = H + 3 - ks, W + 3 - ks
    x, xr = bt(nhwc(rnd(B, Cin, H, W, seed=40)), split)
    wt = rnd(Cout, Cin, ks, ks, seed=41, scale=(ks * ks * Cin) ** -0.5)
    w, wr = bt(wt.permute(0, 2, 3, 1).reshape(Cout, ks * ks, Cin).contiguous(), split)
    wr_t = wr.reshape(Cout, ks, ks, Cin).permute(0, 3, 1, 2)
    bias = rnd(Cout, seed=42)
    y = BT.empty((B, Ho, Wo, Cout), split, DEV)
    ops.conv_fwd(x, w, bias.to(DEV), y, B, H, W, Cin, Cout, ks)
    xin = xr.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wv = wr_t.clone().requires_grad_(True)
    ref = F.conv2d(xin, wv, bias.double(), padding=1)
    assert ref.shape[-2:] == (Ho, Wo)
    close(y.float(), nhwc(ref.detach()), tol_out(split), what="conv kxk fwd")
    dy, dyr = bt(nhwc(rnd(B, Cout, Ho, Wo, seed=43)), split)
    gx, gw = torch.autograd.grad((ref * dyr.permute(0, 3, 1, 2)).sum(), [xin, wv])
    dx = BT.empty((B, H, W, Cin), split, DEV)
    ops.conv_dgrad(dy, w, dx, B, H, W, Cin, Cout, ks)
    close(dx.float(), nhwc(gx), tol_out(split), what="conv kxk dgrad")
    dw, db = torch.zeros(Cout, ks * ks, Cin, device=DEV), torch.full((Cout,), 0.25, device=DEV)
    ops.conv_wgrad(dy, x, dw, B, H, W, Cin, Cout, ks, dbias=db)
    close(dw, gw.permute(0, 2, 3, 1).reshape(Cout, ks * ks, Cin), 3e-5, what="conv kxk wgrad")
    close(db - 0.25, dyr.sum((0, 1, 2)), 3e-5, what="conv kxk bias gradient")


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 14, 48, 24), (1, 7, 256, 128), (2, 5, 16, 8), (2, 20, 96, 48), (1, 33, 96, 48), (3, 4, 96, 48), (1, 1, 96, 48), (2, 8, 96, 48), (1, 17, 96, 48)])
def test_convT(split, B, H, Cin, Cout):  # 96 -> 48 unsplit runs the direct sub-pixel kernel (conv_direct.hip)
    W = H + 1
    x, xr = bt(nhwc(rnd(B, Cin, H, W, seed=30)), split)
    wt = rnd(Cin, Cout, 3, 3, seed=31, scale=(2.25 * Cin) ** -0.5)  # torch ConvTranspose2d layout (Cin,Cout,kh,kw)
    w, wr = bt(wt.permute(1, 2, 3, 0).reshape(Cout, 9, Cin).contiguous(), split)  # Wc[co][tap][ci]
    wr_t = wr.reshape(Cout, 3, 3, Cin).permute(3, 0, 1, 2)
    bias = rnd(Cout, seed=32)
    y = BT.empty((B, 2 * H, 2 * W, Cout), split, DEV)
    ops.convT_fwd(x, w, bias.to(DEV), y, B, H, W, Cin, Cout)
    xin = xr.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wv = wr_t.clone().requires_grad_(True)
    ref = F.conv_transpose2d(xin, wv, bias.double(), stride=2, padding=1, output_padding=1)
    close(y.float(), nhwc(ref.detach()), tol_out(split), what="convT fwd")
    dy, dyr = bt(nhwc(rnd(B, Cout, 2 * H, 2 * W, seed=33)), split)
    gx, gw = torch.autograd.grad((ref * dyr.permute(0, 3, 1, 2)).sum(), [xin, wv])
    dx = BT.empty((B, H, W, Cin), split, DEV)
    ops.convT_dgrad(dy, w, dx, B, H, W, Cin, Cout)
    close(dx.float(), nhwc(gx), tol_out(split), what="convT dgrad")
    dw, db = torch.zeros(Cout, 9, Cin, device=DEV), torch.full((Cout,), 0.25, device=DEV)
    ops.convT_wgrad(dy, x, dw, B, H, W, Cin, Cout, dbias=db)
    close(dw, gw.permute(1, 2, 3, 0).reshape(Cout, 9, Cin), 3e-5, what="convT wgrad")
    close(db - 0.25, dyr.sum((0, 1, 2)), 3e-5, what="convT bias gradient")


def test_two_streams_do_not_share_scratch(monkeypatch):
    """The library's scratch buffers are keyed by (device, stream, slot) (runtime.hip): conv8 re-packs its weights into slot 1 on every call
    and the GEMM behind it reads them, and the BatchNorm statistics kernels keep their workgroup partials in slot 0.  Two streams of one
    device -- a distillation teacher's forward beside the student's step (segmentation.py:216-451) -- used to share ONE buffer per slot and
    could overwrite each other silently.  Here two streams run the same convolution with DIFFERENT weights, interleaved
    launch by launch: every result must equal its single-stream result bit for bit (the model-level form -- a teacher forward beside a
    student step -- is tests/test_gpu_pipeline.py::test_teacher_forward_on_a_side_stream_beside_a_student_step)."""
    monkeypatch.setenv("IG_CONV8", "2")
    monkeypatch.setenv("IG_CONV_DIRECT", "0")
    B, H, Cin, Cout = 2, 24, 128, 256
    W = H + 3
    x, _ = bt(nhwc(rnd(B, Cin, H, W, seed=71)), False)
    ws = [bt((rnd(Cout, Cin, 3, 3, seed=72 + i, scale=(9 * Cin) ** -0.5)).permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous(), False)[0] for i in range(2)]
    bias = rnd(Cout, seed=75).to(DEV)
    want = []
    for w in ws:
        y = BT.zeros((B, H, W, Cout), False, DEV)
        ops.conv3x3_fwd(x, w, bias, y, B, H, W, Cin, Cout)
        assert ops.last_kernel().startswith("conv8_kernel"), ops.last_kernel()
        want.append(y.hi.clone())
    assert not torch.equal(want[0], want[1])
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[BT.zeros((B, H, W, Cout), False, DEV) for _ in range(12)] for _ in range(2)]
    for it in range(12):
        for si in range(2):
            with torch.cuda.stream(streams[si]):
                ops.conv3x3_fwd(x, ws[si], bias, outs[si][it], B, H, W, Cin, Cout)
    torch.cuda.synchronize()
    for si in range(2):
        for it in range(12):
            assert torch.equal(outs[si][it].hi, want[si]), f"stream {si}, launch {it}: conv8 read the other stream's packed weights"


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("B,H,Cin,Cout", [
    (3, 14, 384, 384),   # 256 x 192 instance (N = 384 = 2 x 192), ragged last row tile
    (2, 9, 192, 192),    # N = 192, K = 27 K-tiles padded to 28
    (1, 12, 128, 256),   # 256 x 256 instance, K = 9 x 128
    (2, 10, 128, 128),   # 256 x 128 instance
    (1, 20, 144, 144),   # 192 x 144 instance (T = 3 widths: 2.25 K-tiles per tap, chunk-level tap decode)
    (1, 14, 288, 288),   # 2 x 144
    (2, 11, 96, 192),    # 1.5 K-tiles per tap; data gradient N = 96 on a ragged 128-wide tile
    (1, 7, 576, 576),    # 3 x 192
    (1, 5, 1152, 1152),  # 6 x 192, K = 162 K-tiles, 1296-entry chunk table
    (1, 1, 192, 192), (1, 2, 384, 192),
])
def test_conv8_engine_conv3x3(split, B, H, Cin, Cout, monkeypatch):
    """The 8-phase implicit-GEMM engine with gathering LDS-DMA (conv8.hip), forced for every covered shape: nn.Conv2d(k=3, padding=1)
    forward (+ bias, + eval-mode BatchNorm/ReLU fold) and data gradient (+ the dropout mask of its input) against float64 torch on the
    same rounded operands; every tile instance; repeated launches bit-identical (LDS-DMA / barrier race screen)."""
    monkeypatch.setenv("IG_CONV8", "2")
    monkeypatch.setenv("IG_CONV_DIRECT", "0")  # the 96-channel stage has a direct kernel that would be tried first
    W = H + 3
    x, xr = bt(nhwc(rnd(B, Cin, H, W, seed=26)), split)
    wt = rnd(Cout, Cin, 3, 3, seed=27, scale=(9 * Cin) ** -0.5)
    w, wr = bt(wt.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous(), split)
    wr_t = wr.reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    bias = rnd(Cout, seed=28)
    y = BT.zeros((B, H, W, Cout), split, DEV)
    ops.conv3x3_fwd(x, w, bias.to(DEV), y, B, H, W, Cin, Cout)
    assert ops.last_kernel().startswith("conv8_kernel"), ops.last_kernel()
    xin = xr.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wv = wr_t.clone().requires_grad_(True)
    ref = F.conv2d(xin, wv, bias.double(), padding=1)
    close(y.float(), nhwc(ref.detach()), tol_out(split), what="conv8 fwd")
    first = y.hi.clone()
    for _ in range(5):
        y.hi.zero_()
        ops.conv3x3_fwd(x, w, bias.to(DEV), y, B, H, W, Cin, Cout)
        assert torch.equal(y.hi, first), "conv8 forward differs between identical launches"
    # eval-mode BatchNorm + ReLU folded into the epilogue
    sc, sh = (rnd(Cout, seed=61).abs() + 0.5), rnd(Cout, seed=62)
    ops.conv3x3_fwd(x, w, bias.to(DEV), y, B, H, W, Cin, Cout, bn_scale=sc.to(DEV), bn_shift=sh.to(DEV))
    assert ops.last_kernel().startswith("conv8_kernel")
    refbn = torch.relu(ref.detach() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    close(y.float(), nhwc(refbn), tol_out(split), what="conv8 fwd + BN/ReLU fold")
    dy, dyr = bt(nhwc(rnd(B, Cout, H, W, seed=29)), split)
    (gx,) = torch.autograd.grad((ref * dyr.permute(0, 3, 1, 2)).sum(), [xin])
    dx = BT.zeros((B, H, W, Cin), split, DEV)
    ops.conv3x3_dgrad(dy, w, dx, B, H, W, Cin, Cout)
    assert ops.last_kernel().startswith("conv8_kernel"), ops.last_kernel()
    close(dx.float(), nhwc(gx), tol_out(split), what="conv8 dgrad")
    # dropout mask of the conv input: the same counter hash as the round-1 engine (zero pattern identical, kept values scaled)
    dxd = BT.zeros((B, H, W, Cin), split, DEV)
    ops.conv3x3_dgrad(dy, w, dxd, B, H, W, Cin, Cout, seed=1234, p=0.1)
    assert ops.last_kernel().startswith("conv8_kernel")
    monkeypatch.setenv("IG_CONV8", "0")
    monkeypatch.setenv("IG_CONV_DIRECT", "0")
    dxo = BT.zeros((B, H, W, Cin), split, DEV)
    ops.conv3x3_dgrad(dy, w, dxo, B, H, W, Cin, Cout, seed=1234, p=0.1)
    assert not ops.last_kernel().startswith("conv8_kernel")
    assert torch.equal(dxd.float() == 0, dxo.float() == 0), "dropout masks of the two engines differ"
    close(dxd.float(), dxo.float().double().cpu(), tol_out(split), what="conv8 dgrad with dropout vs the gather GEMM")


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("kind,B,H,Cin,Cout", [
    ("conv", 3, 14, 384, 384), ("conv", 2, 9, 192, 192), ("conv", 1, 7, 576, 576), ("conv", 1, 5, 1152, 1152), ("conv", 1, 1, 192, 192),
    ("conv", 1, 2, 384, 192), ("conv", 2, 28, 384, 384),
    ("convT", 3, 14, 768, 384), ("convT", 2, 9, 384, 192), ("convT", 1, 13, 192, 96), ("convT", 1, 6, 2304, 1152), ("convT", 1, 1, 384, 192), ("convT", 2, 28, 384, 192),
    ("conv", 2, 11, 96, 288), ("conv", 1, 14, 288, 288), ("conv", 1, 9, 96, 96), ("convT", 1, 13, 192, 96), ("convT", 1, 7, 576, 288),  # 256 x 96 tiles
    ("conv", 1, 20, 144, 144), ("convT", 1, 7, 288, 144), ("conv", 2, 13, 288, 144),  # N = 144: one ragged 192-wide tile
])
def test_conv4_equals_conv8_bit_for_bit(split, kind, B, H, Cin, Cout, monkeypatch):
    """conv4_kernel (conv8.hip: the 4-wave 256 x 192 form with a generated K-loop, A pieces gathered with three vector instructions of address
    arithmetic each) against conv8_kernel on the same packed weights and chunk table: Conv2d forward (+ bias, + eval BatchNorm / ReLU fold),
    its data gradient (+ dropout mask), the ConvTranspose data gradient and forward (+ bias, + dropout), widths that tile by 192 (256 x 192
    tiles) or by 96 (256 x 96), plain and the paired split form.  Same MFMA instruction in the same K (and hi / lo product)
    order, so the results must be IDENTICAL; the float64 comparison of conv8 itself is test_conv8_engine_*."""
    monkeypatch.setenv("IG_CONV8", "2")
    monkeypatch.setenv("IG_CONV_DIRECT", "0")
    W = H + 3
    outs = {}
    for arm in ("0", "2"):
        monkeypatch.setenv("IG_GEMM4", arm)
        want = "conv4_kernel" if arm == "2" else "conv8_kernel"
        got = []
        if kind == "conv":
            x, _ = bt(nhwc(rnd(B, Cin, H, W, seed=26)), split)
            w, _ = bt(rnd(Cout, Cin, 3, 3, seed=27, scale=(9 * Cin) ** -0.5).permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous(), split)
            bias = rnd(Cout, seed=28).to(DEV)
            y = BT.zeros((B, H, W, Cout), split, DEV)
            if Cout % 96 == 0 or Cout == 144:
                ops.conv3x3_fwd(x, w, bias, y, B, H, W, Cin, Cout)
                assert ops.last_kernel().startswith(want), ops.last_kernel()
                got.append(y.hi.clone())
                if split:
                    got.append(y.lo.clone())
                sc, sh = (rnd(Cout, seed=61).abs() + 0.5).to(DEV), rnd(Cout, seed=62).to(DEV)
                ops.conv3x3_fwd(x, w, bias, y, B, H, W, Cin, Cout, bn_scale=sc, bn_shift=sh)
                assert ops.last_kernel().startswith(want), ops.last_kernel()
                got.append(y.hi.clone())
                if split:
                    got.append(y.lo.clone())
            if Cin % 96 == 0 or Cin == 144:
                dy, _ = bt(nhwc(rnd(B, Cout, H, W, seed=29)), split)
                dx = BT.zeros((B, H, W, Cin), split, DEV)
                ops.conv3x3_dgrad(dy, w, dx, B, H, W, Cin, Cout)
                assert ops.last_kernel().startswith(want), ops.last_kernel()
                got.append(dx.hi.clone())
                if split:
                    got.append(dx.lo.clone())
                ops.conv3x3_dgrad(dy, w, dx, B, H, W, Cin, Cout, seed=1234, p=0.1)
                assert ops.last_kernel().startswith(want), ops.last_kernel()
                got.append(dx.hi.clone())
                if split:
                    got.append(dx.lo.clone())
        else:
            w, _ = bt(rnd(Cin, Cout, 3, 3, seed=31, scale=(2.25 * Cin) ** -0.5).permute(1, 2, 3, 0).reshape(Cout, 9, Cin).contiguous(), split)
            dy, _ = bt(nhwc(rnd(B, Cout, 2 * H, 2 * W, seed=33)), split)
            dx = BT.zeros((B, H, W, Cin), split, DEV)
            ops.convT_dgrad(dy, w, dx, B, H, W, Cin, Cout)
            assert ops.last_kernel().startswith(want), ops.last_kernel()
            got.append(dx.hi.clone())
            if split:
                got.append(dx.lo.clone())
            if Cout % 96 == 0 or Cout == 144:  # forward: four sub-pixel phases (own tables, K lengths and row pitches) as tiles of one launch
                x, _ = bt(nhwc(rnd(B, Cin, H, W, seed=30)), split)
                bias = rnd(Cout, seed=32).to(DEV)
                y = BT.zeros((B, 2 * H, 2 * W, Cout), split, DEV)
                ops.convT_fwd(x, w, bias, y, B, H, W, Cin, Cout)
                assert ops.last_kernel().startswith(want), ops.last_kernel()
                got.append(y.hi.clone())
                if split:
                    got.append(y.lo.clone())
                y.hi.zero_()
                ops.convT_fwd(x, w, bias, y, B, H, W, Cin, Cout, seed=77, p=0.1)
                assert ops.last_kernel().startswith(want), ops.last_kernel()
                got.append(y.hi.clone())
                if split:
                    got.append(y.lo.clone())
        assert got
        outs[arm] = got
    for i, (a, b) in enumerate(zip(outs["0"], outs["2"])):
        assert torch.equal(a, b), f"conv4 differs from conv8 (output {i}): max {((a.float() - b.float()).abs().max().item()):.3e}"


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("B,H,Cin,Cout", [
    (3, 14, 768, 384),   # stage 0 of the 100M head: N = 384 (fwd), 768 (dgrad = 3 x 256)
    (2, 9, 384, 192),    # stage 1
    (1, 13, 192, 96),    # stage 2: 1-tap phase = 3 K-tiles padded to 4; N = 96 instance
    (1, 7, 288, 144),    # T = 3 widths
    (1, 6, 2304, 1152),  # T = 3 stage 0
    (2, 5, 128, 128), (1, 1, 384, 192), (1, 2, 128, 256),
])
def test_conv8_engine_convT(split, B, H, Cin, Cout, monkeypatch):
    """nn.ConvTranspose2d(k3, s2, p1, op1) forward (four sub-pixel phases as tiles of one persistent launch, + bias, + dropout)
    and data gradient (stride-2 gather) on the conv8 engine against float64 torch."""
    monkeypatch.setenv("IG_CONV8", "2")
    monkeypatch.setenv("IG_CONV_DIRECT", "0")
    W = H + 1
    x, xr = bt(nhwc(rnd(B, Cin, H, W, seed=30)), split)
    wt = rnd(Cin, Cout, 3, 3, seed=31, scale=(2.25 * Cin) ** -0.5)
    w, wr = bt(wt.permute(1, 2, 3, 0).reshape(Cout, 9, Cin).contiguous(), split)
    wr_t = wr.reshape(Cout, 3, 3, Cin).permute(3, 0, 1, 2)
    bias = rnd(Cout, seed=32)
    y = BT.zeros((B, 2 * H, 2 * W, Cout), split, DEV)
    ops.convT_fwd(x, w, bias.to(DEV), y, B, H, W, Cin, Cout)
    assert ops.last_kernel().startswith("conv8_kernel"), ops.last_kernel()
    xin = xr.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wv = wr_t.clone().requires_grad_(True)
    ref = F.conv_transpose2d(xin, wv, bias.double(), stride=2, padding=1, output_padding=1)
    close(y.float(), nhwc(ref.detach()), tol_out(split), what="conv8 convT fwd")
    first = y.hi.clone()
    for _ in range(5):
        y.hi.zero_()
        ops.convT_fwd(x, w, bias.to(DEV), y, B, H, W, Cin, Cout)
        assert torch.equal(y.hi, first), "conv8 convT forward differs between identical launches"
    dy, dyr = bt(nhwc(rnd(B, Cout, 2 * H, 2 * W, seed=33)), split)
    (gx,) = torch.autograd.grad((ref * dyr.permute(0, 3, 1, 2)).sum(), [xin])
    dx = BT.zeros((B, H, W, Cin), split, DEV)
    ops.convT_dgrad(dy, w, dx, B, H, W, Cin, Cout)
    assert ops.last_kernel().startswith("conv8_kernel"), ops.last_kernel()
    close(dx.float(), nhwc(gx), tol_out(split), what="conv8 convT dgrad")
    # forward dropout: same mask as the round-1 engine
    yd = BT.zeros((B, 2 * H, 2 * W, Cout), split, DEV)
    ops.convT_fwd(x, w, bias.to(DEV), yd, B, H, W, Cin, Cout, seed=77, p=0.1)
    monkeypatch.setenv("IG_CONV8", "0")
    monkeypatch.setenv("IG_CONV_DIRECT", "0")
    yo = BT.zeros((B, 2 * H, 2 * W, Cout), split, DEV)
    ops.convT_fwd(x, w, bias.to(DEV), yo, B, H, W, Cin, Cout, seed=77, p=0.1)
    assert not ops.last_kernel().startswith("conv8_kernel")
    assert torch.equal(yd.float() == 0, yo.float() == 0), "dropout masks of the two engines differ"
    close(yd.float(), yo.float().double().cpu(), tol_out(split), what="conv8 convT fwd with dropout vs the gather GEMM")


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("kind,B,H,Cin,Cout", [
    ("conv", 3, 14, 384, 384),    # 2 x 192 rows, 3456 = 18 x 192 columns; ragged token tail
    ("conv", 2, 9, 192, 192),     # one row tile, a tap every 1.5 half-tiles (lane-level tap decode)
    ("conv", 1, 20, 144, 144),    # ragged rows (144 of 192) and a ragged last column tile (1296 = 6.75 x 192)
    ("conv", 1, 14, 288, 288),    # 1.5 row tiles
    ("conv", 2, 11, 96, 96),      # half a row tile (forced)
    ("conv", 1, 7, 576, 576), ("conv", 1, 12, 128, 256), ("conv", 1, 1, 192, 192), ("conv", 1, 3, 48, 192),
    ("convT", 3, 14, 768, 384),   # 9 taps x 2 x 3 tiles of 192 x 256
    ("convT", 2, 9, 384, 192), ("convT", 1, 13, 192, 96), ("convT", 1, 7, 288, 144), ("convT", 1, 6, 2304, 1152), ("convT", 2, 5, 128, 128),
    ("convT", 1, 1, 384, 192),
])
def test_wgrad8_conv_engine(split, kind, B, H, Cin, Cout, monkeypatch):
    """Weight gradients of nn.Conv2d(k=3, padding=1) / nn.ConvTranspose2d(k3, s2, p1, op1) on the grouped 8-phase engine with a
    gathering LDS-DMA operand (gemm8w.hip modes 1 / 2), forced for every shape: against float64 autograd on the same rounded
    operands, accumulation into dW, the bias gradient, bit-identical repeats."""
    monkeypatch.setenv("IG_WGRAD8_CONV", "2")
    W = H + 2
    x, xr = bt(nhwc(rnd(B, Cin, H, W, seed=50)), split)
    xin = xr.permute(0, 3, 1, 2).clone()
    if kind == "conv":
        wv = rnd(Cout, Cin, 3, 3, seed=51).double().requires_grad_(True)
        ref = F.conv2d(xin, wv, None, padding=1)
        Ho, Wo = H, W
    else:
        wv = rnd(Cin, Cout, 3, 3, seed=51).double().requires_grad_(True)
        ref = F.conv_transpose2d(xin, wv, None, stride=2, padding=1, output_padding=1)
        Ho, Wo = 2 * H, 2 * W
    dy, dyr = bt(nhwc(rnd(B, Cout, Ho, Wo, seed=52)), split)
    (gw,) = torch.autograd.grad((ref * dyr.permute(0, 3, 1, 2)).sum(), [wv])
    gw_s = (gw.permute(0, 2, 3, 1) if kind == "conv" else gw.permute(1, 2, 3, 0)).reshape(Cout, 9, Cin)
    dw, db = torch.zeros(Cout, 9, Cin, device=DEV), torch.full((Cout,), 0.25, device=DEV)
    fn = ops.conv3x3_wgrad if kind == "conv" else ops.convT_wgrad
    fn(dy, x, dw, B, H, W, Cin, Cout, dbias=db)
    close(dw, gw_s, 3e-5, what=f"{kind} wgrad8")
    close(db - 0.25, dyr.sum((0, 1, 2)), 3e-5, what=f"{kind} wgrad8 bias gradient")
    first = dw.clone()
    fn(dy, x, dw, B, H, W, Cin, Cout)  # accumulates
    assert ops.last_kernel().startswith("gemm8w_kernel"), ops.last_kernel()
    close(dw, 2 * gw_s, 3e-5, what=f"{kind} wgrad8 accumulate")
    for _ in range(3):
        dw.zero_()
        fn(dy, x, dw, B, H, W, Cin, Cout)
        assert torch.equal(dw, first), "wgrad8 differs between identical launches"


@pytest.mark.parametrize("kind,C,H", [("conv", 48, 224), ("conv", 96, 112), ("convT", 96, 112)])
def test_direct_head_kernels_at_model_size_vs_float64(kind, C, H):
    """The direct (halo-tile / sub-pixel) head kernels at the BASELINE image sizes -- 224 x 224 x 48, 112 x 112 x 96 and the
    96 -> 48 ConvTranspose 112 -> 224 -- against float64 torch on the SAME bf16-rounded operands (forward, data gradient, weight
    and bias gradient; full tensors, not samples).  tests/test_gpu_direct_vs_gemm.py only compares two HIP paths with each other."""
    B = 2
    if kind == "conv":
        Cin = Cout = C
        x, xr = bt(nhwc(rnd(B, Cin, H, H, seed=70)), False)
        wt = rnd(Cout, Cin, 3, 3, seed=71, scale=(9 * Cin) ** -0.5)
        w, wr = bt(wt.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous(), False)
        bias = rnd(Cout, seed=72)
        y = BT.empty((B, H, H, Cout), False, DEV)
        ops.conv3x3_fwd(x, w, bias.to(DEV), y, B, H, H, Cin, Cout)
        xin = xr.permute(0, 3, 1, 2).clone().requires_grad_(True)
        wv = wr.reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2).clone().requires_grad_(True)
        ref = F.conv2d(xin, wv, bias.double(), padding=1)
        Ho = H
    else:
        Cin, Cout = C, C // 2
        x, xr = bt(nhwc(rnd(B, Cin, H, H, seed=73)), False)
        wt = rnd(Cin, Cout, 3, 3, seed=74, scale=(9 * Cin) ** -0.5)  # nn.ConvTranspose2d weight (Cin, Cout, 3, 3)
        w, wr = bt(wt.permute(1, 2, 3, 0).reshape(Cout, 9, Cin).contiguous(), False)  # stored Wc[Cout][9][Cin]
        bias = rnd(Cout, seed=75)
        y = BT.empty((B, 2 * H, 2 * H, Cout), False, DEV)
        ops.convT_fwd(x, w, bias.to(DEV), y, B, H, H, Cin, Cout)
        xin = xr.permute(0, 3, 1, 2).clone().requires_grad_(True)
        wv = wr.reshape(Cout, 3, 3, Cin).permute(3, 0, 1, 2).clone().requires_grad_(True)
        ref = F.conv_transpose2d(xin, wv, bias.double(), stride=2, padding=1, output_padding=1)
        Ho = 2 * H
    close(y.float(), nhwc(ref.detach()), tol_out(False), what=f"{kind} fwd {C}ch {H}px")
    dy, dyr = bt(nhwc(rnd(B, Cout, Ho, Ho, seed=76)), False)
    gx, gw = torch.autograd.grad((ref * dyr.permute(0, 3, 1, 2)).sum(), [xin, wv])
    dx = BT.empty((B, H, H, Cin), False, DEV)
    dw, db = torch.zeros(Cout, 9, Cin, device=DEV), torch.zeros(Cout, device=DEV)
    if kind == "conv":
        ops.conv3x3_dgrad(dy, w, dx, B, H, H, Cin, Cout)
        ops.conv3x3_wgrad(dy, x, dw, B, H, H, Cin, Cout, dbias=db)
        gw_s = gw.permute(0, 2, 3, 1).reshape(Cout, 9, Cin)
    else:
        ops.convT_dgrad(dy, w, dx, B, H, H, Cin, Cout)
        ops.convT_wgrad(dy, x, dw, B, H, H, Cin, Cout, dbias=db)
        gw_s = gw.permute(1, 2, 3, 0).reshape(Cout, 9, Cin)
    close(dx.float(), nhwc(gx), tol_out(False), what=f"{kind} dgrad")
    close(dw, gw_s, 3e-5, what=f"{kind} wgrad")
    close(db, dyr.sum((0, 1, 2)), 3e-5, what=f"{kind} bias gradient")


def test_split_direct_conv48_and_its_fused_statistics():
    """The split-precision (bf16x3) form of the 48-channel stage (conv_direct.hip conv3x3_direct_split_kernel: hi + lo images of weights and
    halo in LDS): forward with eval-mode BatchNorm + ReLU folded in, data gradient with the dropout mask, and the training-mode form whose
    epilogue leaves the BatchNorm sums of the STORED hi + lo values; ragged tile edges; float64 references at the split tolerance."""
    C, B, H, W = 48, 3, 21, 37
    x, xr = bt(nhwc(rnd(B, C, H, W, seed=70)), True)
    w, wr = bt(rnd(C, C, 3, 3, seed=71, scale=(9 * C) ** -0.5).permute(0, 2, 3, 1).reshape(C, 9, C).contiguous(), True)
    bias, sc, sh = rnd(C, seed=72), 1 + 0.2 * rnd(C, seed=73), 0.3 * rnd(C, seed=74)
    wt = wr.reshape(C, 3, 3, C).permute(0, 3, 1, 2)
    conv = F.conv2d(xr.permute(0, 3, 1, 2), wt, bias.double(), padding=1)
    y = BT.empty((B, H, W, C), True, DEV)
    ops.conv3x3_fwd(x, w, bias.to(DEV), y, B, H, W, C, C, bn_scale=sc.float().to(DEV), bn_shift=sh.float().to(DEV))
    assert ops.last_kernel().startswith("conv3x3_direct_split_kernel"), ops.last_kernel()
    ref = torch.relu(conv * sc.double().view(1, C, 1, 1) + sh.double().view(1, C, 1, 1))
    close(y.float(), nhwc(ref), tol_out(True), what="split direct conv + bn fold")
    # training form: plain convolution + the statistics of what was stored
    sums = torch.full((2 * C,), -1.0, dtype=torch.float64, device=DEV)
    assert ops.conv3x3_fwd_stats(x, w, bias.to(DEV), y, sums, B, H, W, C, C)
    assert ops.last_kernel().startswith("conv3x3_direct_split_kernel"), ops.last_kernel()
    close(y.float(), nhwc(conv), tol_out(True), what="split direct conv (statistics form)")
    yd = y.float().double().cpu().reshape(-1, C)
    close(sums[:C], yd.sum(0), 2e-6, what="sum")
    close(sums[C:], (yd * yd).sum(0), 2e-6, what="sum of squares")
    sums2 = torch.empty_like(sums)
    ops.conv3x3_fwd_stats(x, w, bias.to(DEV), y, sums2, B, H, W, C, C)
    assert torch.equal(sums, sums2)  # ordered partial sums
    # data gradient, with and without the dropout mask of the convolution's input (the mask is the plain kernels' mask)
    dy, dyr = bt(nhwc(rnd(B, C, H, W, seed=75)), True)
    gx = F.conv_transpose2d(dyr.permute(0, 3, 1, 2), wt, padding=1)
    dx = BT.empty((B, H, W, C), True, DEV)
    ops.conv3x3_dgrad(dy, w, dx, B, H, W, C, C)
    assert ops.last_kernel().startswith("conv3x3_direct_split_kernel"), ops.last_kernel()
    close(dx.float(), nhwc(gx), tol_out(True), what="split direct dgrad")
    dxm, dxp = BT.empty((B, H, W, C), True, DEV), BT.empty((B, H, W, C), False, DEV)
    ops.conv3x3_dgrad(dy, w, dxm, B, H, W, C, C, seed=99, p=0.1)
    ops.conv3x3_dgrad(BT(dy.hi), BT(w.hi), dxp, B, H, W, C, C, seed=99, p=0.1)
    assert ((dxm.float() == 0) != (dxp.float() == 0)).float().mean().item() < 1e-4  # same counter-based mask as the plain kernel
    close(dxm.float(), nhwc(gx) * (dxm.float() != 0).double().cpu() / 0.9, tol_out(True), what="split direct dgrad x dropout mask")


@pytest.mark.parametrize("lo_first", [False, True])
def test_split_three_pass_fallback(lo_first, monkeypatch):
    """C-ABI callers may hand over hi and lo buffers that are NOT one block (lo below hi, or far away): gemm8 / conv8 then run the
    three-pass form of the split mode (NSEG = 3) instead of the paired K-tiles (NSEG = 2, what ops.BT's allocations get).  Both forms
    against float64, and against each other."""
    monkeypatch.setenv("IG_GEMM8", "2")
    monkeypatch.setenv("IG_CONV8", "2")
    M, N, K = 1300, 512, 256

    def separate(t: BT) -> BT:  # the same values in two separate allocations, lo below hi when asked
        a, b = torch.empty_like(t.hi), torch.empty_like(t.hi)
        if (b.data_ptr() < a.data_ptr()) != lo_first:
            a, b = b, a
        a.copy_(t.hi), b.copy_(t.lo)
        return BT(a, b)

    x, xr = bt(rnd(M, K, seed=1), True)
    w, wr = bt(rnd(N, K, seed=2, scale=K**-0.5), True)
    b = rnd(N, seed=3).to(DEV)
    ref = xr @ wr.t() + b.double().cpu()
    y2, y3 = BT.zeros((M, N), True, DEV), BT.zeros((M, N), True, DEV)
    ops.linear_fwd(x, w, b, y2, M, N, K, act=0)
    assert ops.last_kernel().startswith("gemm8_kernel<0,2,"), ops.last_kernel()
    xs, ws = separate(x), separate(w)
    ops.linear_fwd(xs, ws, b, y3, M, N, K, act=0)
    if lo_first:
        assert ops.last_kernel().startswith("gemm8_kernel<0,3,"), ops.last_kernel()
    close(y2.float(), ref, tol_out(True), what="paired")
    close(y3.float(), ref, tol_out(True), what="three-pass (or paired across two allocations)")
    monkeypatch.setenv("IG_G8_PAIR", "0")
    ops.linear_fwd(x, w, b, y3, M, N, K, act=0)
    assert ops.last_kernel().startswith("gemm8_kernel<0,3,"), ops.last_kernel()
    close(y3.float(), ref, tol_out(True), what="three-pass")
    close(y3.float(), y2.float().double().cpu(), tol_out(True), what="three-pass vs paired")
    # conv8, gathered operand
    B, H, C = 2, 16, 256
    xi, xir = bt(nhwc(rnd(B, C, H, H, seed=5)), True)
    wc, wcr = bt(rnd(C, C, 3, 3, seed=6, scale=(9 * C) ** -0.5).permute(0, 2, 3, 1).reshape(C, 9, C).contiguous(), True)
    cref = nhwc(F.conv2d(xir.permute(0, 3, 1, 2), wcr.reshape(C, 3, 3, C).permute(0, 3, 1, 2), None, padding=1))
    yc3, yc2 = BT.empty((B, H, H, C), True, DEV), BT.empty((B, H, H, C), True, DEV)
    ops.conv3x3_fwd(xi, wc, None, yc3, B, H, H, C, C)
    assert ops.last_kernel().startswith("conv8_kernel") and ",3,true>" in ops.last_kernel(), ops.last_kernel()
    monkeypatch.delenv("IG_G8_PAIR")
    ops.conv3x3_fwd(xi, wc, None, yc2, B, H, H, C, C)
    assert ops.last_kernel().startswith("conv8_kernel") and ",2,true>" in ops.last_kernel(), ops.last_kernel()
    close(yc2.float(), cref, tol_out(True), what="conv8 paired")
    close(yc3.float(), cref, tol_out(True), what="conv8 three-pass")
    ops.conv3x3_fwd(separate(xi), wc, None, yc3, B, H, H, C, C)  # (lo below hi: no common buffer descriptor -> three passes)
    if lo_first:
        assert ",3,true>" in ops.last_kernel(), ops.last_kernel()
    close(yc3.float(), cref, tol_out(True), what="conv8 with separate hi / lo allocations")


@pytest.mark.parametrize("C", [48, 96])
def test_conv3x3_direct_bn_fold(C):
    """C -> C forward with the eval-mode BatchNorm + ReLU folded into the epilogue (direct kernels), ragged tile edges."""
    B, H, W = 2, 19, 35
    x, xr = bt(nhwc(rnd(B, C, H, W, seed=60)), False)
    w, wr = bt(rnd(C, C, 3, 3, seed=61, scale=(9 * C) ** -0.5).permute(0, 2, 3, 1).reshape(C, 9, C).contiguous(), False)
    bias, sc, sh = rnd(C, seed=62), 1 + 0.2 * rnd(C, seed=63), 0.3 * rnd(C, seed=64)
    y = BT.empty((B, H, W, C), False, DEV)
    ops.conv3x3_fwd(x, w, bias.to(DEV), y, B, H, W, C, C, bn_scale=sc.float().to(DEV), bn_shift=sh.float().to(DEV))
    ref = F.conv2d(xr.permute(0, 3, 1, 2), wr.reshape(C, 3, 3, C).permute(0, 3, 1, 2), bias.double(), padding=1)
    ref = torch.relu(ref * sc.double().view(1, C, 1, 1).to(ref.device) + sh.double().view(1, C, 1, 1).to(ref.device))
    close(y.float(), nhwc(ref), tol_out(False), what="direct conv + bn fold")


@pytest.mark.parametrize("Cin,Cout", [(8, 16), (8, 48), (96, 48), (8, 96)])
def test_dropout_mask_consistency(Cin, Cout):
    """ConvT forward mask == conv dgrad mask (same counter-based hash; implicit-GEMM and direct kernels), keep rate ~ 1-p."""
    B, H, W = 1, 8, 8
    x = BT.from_float(torch.zeros(B, H, W, Cin, device=DEV), False)
    w = BT.from_float(torch.zeros(Cout, 9, Cin, device=DEV), False)
    y = BT.empty((B, 2 * H, 2 * W, Cout), False, DEV)
    ops.convT_fwd(x, w, torch.ones(Cout, device=DEV), y, B, H, W, Cin, Cout, seed=123, p=0.1)
    mask_f = y.float()  # = mask/0.9 since convT(0)+1
    keep = (mask_f > 0).float().mean().item()
    assert abs(keep - 0.9) < 0.02
    vals = mask_f[mask_f > 0]
    assert torch.allclose(vals, torch.full_like(vals, 1 / 0.9), rtol=1e-2)
    # dgrad of a centre-tap-only identity conv with dy = 1 reproduces the mask on d(conv input)
    C = Cout
    wt = torch.zeros(C, 9, C, device=DEV)
    wt[torch.arange(C), 4, torch.arange(C)] = 1.0
    dy = BT.from_float(torch.ones(B, 2 * H, 2 * W, C, device=DEV), False)
    dx = BT.empty((B, 2 * H, 2 * W, C), False, DEV)
    ops.conv3x3_dgrad(dy, BT.from_float(wt, False), dx, B, 2 * H, 2 * W, C, C, seed=123, p=0.1)
    assert torch.equal(dx.float() > 0, mask_f > 0)


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("M,C", [(2 * 28 * 28, 48), (1000, 192), (70000, 24)])
def test_bn_relu(split, M, C):
    x, xr = bt(rnd(M, C, seed=34) * 1.5 + 0.3, split)
    g = 1 + 0.1 * rnd(C, seed=35)
    b = 0.1 * rnd(C, seed=36)
    rm0, rv0 = 0.1 * rnd(C, seed=37), 1 + 0.2 * torch.rand(C)
    rm, rv = rm0.clone().to(DEV), rv0.clone().to(DEV)
    y = BT.empty((M, C), split, DEV)
    scale, shift, mean, rstd = (torch.empty(C, device=DEV) for _ in range(4))
    sums = torch.empty(2 * C, dtype=torch.float64, device=DEV)
    ops.bn_relu_fwd(x, g.to(DEV), b.to(DEV), rm, rv, y, scale, shift, mean, rstd, sums, M, C, True, True)
    xd = xr.clone().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    rmd, rvd = rm0.double().clone(), rv0.double().clone()
    ref = F.relu(F.batch_norm(xd, rmd, rvd, gd, bd, True, 0.1, 1e-5))
    close(y.float(), ref.detach(), tol_out(split), what="bn fwd")
    close(rm, rmd, 1e-5, what="running mean")
    close(rv, rvd, 1e-5, what="running var")
    dy, dyr = bt(rnd(M, C, seed=38), split)
    gx, gg, gb = torch.autograd.grad((ref * dyr).sum(), [xd, gd, bd])
    dx = BT.empty((M, C), split, DEV)
    dgam, dbet = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ops.bn_relu_bwd(x, dy, scale, shift, mean, rstd, dx, dgam, dbet, sums, M, C)
    close(dx.float(), gx, tol_out(split), what="bn dx")
    close(dgam, gg, 3e-5, what="bn dgamma")
    close(dbet, gb, 3e-5, what="bn dbeta")
    # eval mode uses running statistics
    ops.bn_relu_fwd(x, g.to(DEV), b.to(DEV), rm, rv, y, scale, shift, mean, rstd, sums, M, C, False, False)
    ref_e = F.relu(F.batch_norm(xr, rmd, rvd, g.double(), b.double(), False, 0.1, 1e-5))
    close(y.float(), ref_e, tol_out(split), what="bn eval")


@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("ncls,C", [(2, 48), (13, 144), (7, 96), (3, 48), (13, 16), (7, 8)])  # the last two: narrow heads, > 64 KiB of staged dlogits
def test_classifier_and_loss(split, ncls, C):
    B, H, W = 2, 24, 20
    HW = H * W
    f, fr = bt(rnd(B, HW, C, seed=39), split)
    w = rnd(ncls, C, seed=40, scale=C**-0.5)
    b = rnd(ncls, seed=41)
    logits = torch.empty(B, ncls, H, W, device=DEV)
    ops.classifier_fwd(f, w.to(DEV), b.to(DEV), logits, B, HW, C, ncls)
    fd = fr.clone().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = (fd @ wd.t() + bd).permute(0, 2, 1).reshape(B, ncls, H, W)
    close(logits, ref.detach(), 2e-5, what="classifier")
    g = torch.Generator().manual_seed(5)
    labels = torch.randint(0, ncls, (B, H, W), generator=g)
    labels[torch.rand(B, H, W, generator=g) < 0.1] = -1
    cw = torch.rand(ncls, generator=g) + 0.5
    stats = torch.zeros(2, dtype=torch.float64, device=DEV)
    dlog = torch.empty_like(logits)
    preds = torch.empty(B, H, W, dtype=torch.int64, device=DEV)
    p8 = torch.empty(B, H, W, dtype=torch.int8, device=DEV)
    conf = torch.zeros(ncls, ncls, dtype=torch.int64, device=DEV)
    lg = logits.double().cpu().requires_grad_(True)
    ops.ce_loss(logits, labels.to(DEV), cw.to(DEV), -1, stats, dlog, preds, p8, conf)
    mask = labels.ne(-1)
    per = F.cross_entropy(lg, labels, weight=cw.double(), ignore_index=-1, reduction="none")
    loss_ref = per[mask].mean()  # segmentation.py:120-122
    loss = (stats[0] / stats[1]).item()
    assert stats[1].item() == mask.sum().item()
    assert abs(loss - loss_ref.item()) < 1e-5 * max(1, abs(loss_ref.item()))
    (gl,) = torch.autograd.grad(loss_ref, lg)
    close(dlog / stats[1].float(), gl, 2e-5, what="dlogits")
    pr = lg.detach().argmax(1)
    assert torch.equal(preds.cpu(), pr) and torch.equal(p8.cpu().long(), pr)
    import sys, os
    from oracle import prithvi_oracle as O
    assert np.array_equal(conf.cpu().numpy(), O.confusion_matrix(labels.numpy(), pr.numpy(), ncls, -1))
    assert torch.equal(ops.argmax_i8(logits).cpu().long(), pr)
    conf2 = torch.zeros_like(conf)
    ops.confusion_update(labels.to(DEV).reshape(-1), preds.reshape(-1), conf2, ncls, -1)
    assert torch.equal(conf2, conf)
    # classifier backward with the loss-kernel's un-normalised dlogits + device-side count
    (gf, gw, gb) = torch.autograd.grad((ref * gl).sum(), [fd, wd, bd])
    df = BT.empty((B, HW, C), split, DEV)
    dw = torch.zeros(ncls, C, device=DEV)
    db = torch.zeros(ncls, device=DEV)
    ops.classifier_bwd(dlog, f, w.to(DEV), df, dw, db, stats, B, HW, C, ncls)
    close(df.float(), gf, tol_out(split), what="classifier df")
    close(dw, gw, 3e-5, what="classifier dw")
    close(db, gb, 3e-5, what="classifier db")


@pytest.mark.parametrize("C,B,H,W,expect", [(48, 3, 40, 40, True), (48, 2, 23, 37, True), (48, 70, 32, 32, True), (192, 1, 16, 16, False),
                                            (96, 2, 40, 40, True), (96, 3, 21, 35, True), (96, 40, 32, 32, True)])
def test_conv3x3_fwd_stats(C, B, H, W, expect):
    """nn.Conv2d(k=3, padding=1) in front of a training-mode BatchNorm (model.py:370-377): where the direct 48-channel kernel runs, its
    epilogue also leaves the per-channel sum / sum of squares of the STORED outputs (the statistics pass without its read of the tensor);
    elsewhere the call is the plain convolution and reports that the statistics pass is still owed."""
    x, xr = bt(rnd(B, H, W, C, seed=61), False)
    w, wr = bt(rnd(C, 9, C, seed=62, scale=(9 * C) ** -0.5), False)
    bias = rnd(C, seed=63)
    y, y2 = BT.empty((B, H, W, C), False, DEV), BT.empty((B, H, W, C), False, DEV)
    sums = torch.full((2 * C,), -1.0, dtype=torch.float64, device=DEV)
    fused = ops.conv3x3_fwd_stats(x, w, bias.to(DEV), y, sums, B, H, W, C, C)
    assert fused == expect
    ops.conv3x3_fwd(x, w, bias.to(DEV), y2, B, H, W, C, C)
    assert torch.equal(y.hi, y2.hi)  # the same kernel with the statistics switched on
    ref = F.conv2d(xr.permute(0, 3, 1, 2), wr.view(C, 3, 3, C).permute(0, 3, 1, 2), bias.double(), padding=1).permute(0, 2, 3, 1)
    close(y.float(), ref, tol_out(False), what="conv fwd")
    if fused:
        yd = y.float().double().cpu().reshape(-1, C)
        close(sums[:C], yd.sum(0), 2e-6, what="sum")
        close(sums[C:], (yd * yd).sum(0), 2e-6, what="sum of squares")
        sums2 = torch.empty_like(sums)
        ops.conv3x3_fwd_stats(x, w, bias.to(DEV), y, sums2, B, H, W, C, C)
        assert torch.equal(sums, sums2)  # ordered partial sums: bit-identical from run to run
        # BatchNorm from these sums == the statistics pass over the stored tensor
        g, b = 1 + 0.1 * rnd(C, seed=64), 0.1 * rnd(C, seed=65)
        out = []
        for mode in (0, 1):
            rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            scale, shift, mean, rstd = (torch.empty(C, device=DEV) for _ in range(4))
            if mode:
                ops.bn_finalize(sums, g.to(DEV), b.to(DEV), rm, rv, scale, shift, mean, rstd, B * H * W, C, True)
                # ... and the apply pass from ready statistics == statistics + apply
                ya, yb = BT.empty((B, H, W, C), False, DEV), BT.empty((B, H, W, C), False, DEV)
                r2, v2 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
                ops.bn_relu_fwd(y, g.to(DEV), b.to(DEV), r2, v2, ya, scale, shift, mean, rstd, sums.clone(), B * H * W, C, True, True, stats_ready=True)
                r3, v3 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
                ops.bn_relu_fwd(y, g.to(DEV), b.to(DEV), r3, v3, yb, scale, shift, mean, rstd, torch.empty_like(sums), B * H * W, C, True, True)
                close(ya.float(), yb.float().double().cpu(), 1e-2, what="apply from ready statistics")
                close(r2, r3.double().cpu(), 2e-6, what="running mean (ready statistics)")
            else:
                s3 = torch.empty_like(sums)
                ops.bn_stats(y, g.to(DEV), b.to(DEV), rm, rv, scale, shift, mean, rstd, s3, B * H * W, C, True)
            out.append((rm, rv, scale, shift, mean, rstd))
        for a_, b_, nm in zip(out[0], out[1], ("running mean", "running var", "scale", "shift", "mean", "rstd")):
            close(b_, a_.double().cpu(), 2e-6, what=nm)


@pytest.mark.parametrize("ncls,C,B,H,W,expect", [(2, 48, 3, 40, 40, True), (1, 48, 2, 23, 37, True), (2, 48, 30, 48, 32, True), (3, 48, 1, 16, 16, False),
                                                 (2, 96, 1, 16, 16, False)])
def test_conv3x3_cls_fwd_inference_tail(ncls, C, B, H, W, expect):
    """Inference tail (model.py:370-377 in eval mode + :389): last Conv2d + folded BatchNorm + ReLU with the 1 x 1 classifier applied in the
    same epilogue, against float64 torch and against the two separate kernels; uncovered shapes report that nothing was computed."""
    x, xr = bt(rnd(B, H, W, C, seed=71), False)
    w, wr = bt(rnd(C, 9, C, seed=72, scale=(9 * C) ** -0.5), False)
    bias, sc, sh = rnd(C, seed=73), 1 + 0.2 * rnd(C, seed=74), 0.3 * rnd(C, seed=75)
    cw, cb = rnd(ncls, C, seed=76, scale=C**-0.5), rnd(ncls, seed=77)
    logits = torch.full((B, ncls, H, W), 7.0, device=DEV)
    fused = ops.conv3x3_cls_fwd(x, w, bias.to(DEV), sc.to(DEV), sh.to(DEV), None, cw.to(DEV), cb.to(DEV), logits, B, H, W, C, ncls)
    assert fused == expect
    if not fused:
        assert (logits == 7.0).all()  # nothing was computed
        return
    conv = F.conv2d(xr.permute(0, 3, 1, 2), wr.view(C, 3, 3, C).permute(0, 3, 1, 2), bias.double(), padding=1)
    act = F.relu(conv * sc.double().view(1, C, 1, 1) + sh.double().view(1, C, 1, 1))
    ref = F.conv2d(act, cw.double().view(ncls, C, 1, 1), cb.double())
    close(logits, ref, 1e-4, what="fused inference tail vs float64")
    # the separate kernels round the activation to bf16 in between; with y given the fused call also stores it, identically
    y, y2 = BT.empty((B, H, W, C), False, DEV), BT.empty((B, H, W, C), False, DEV)
    lg2, lg3 = torch.empty_like(logits), torch.empty_like(logits)
    ops.conv3x3_fwd(x, w, bias.to(DEV), y, B, H, W, C, C, sc.to(DEV), sh.to(DEV))
    ops.classifier_fwd(y, cw.to(DEV), cb.to(DEV), lg2, B, H * W, C, ncls)
    close(logits, lg2.double().cpu(), 4e-3, what="fused vs separate kernels")
    assert ops.conv3x3_cls_fwd(x, w, bias.to(DEV), sc.to(DEV), sh.to(DEV), y2, cw.to(DEV), cb.to(DEV), lg3, B, H, W, C, ncls)
    assert torch.equal(y2.hi, y.hi) and torch.equal(lg3, logits)


@pytest.mark.parametrize("det", [False, True])
@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("ncls,C,B,H,W", [(2, 48, 2, 24, 20), (13, 144, 2, 24, 20), (1, 48, 1, 30, 30), (4, 144, 3, 56, 56), (2, 96, 1, 17, 9), (6, 48, 2, 24, 20), (16, 80, 1, 40, 33)])
def test_classifier_bn_fused_tail(det, split, ncls, C, B, H, W):
    """The training-mode tail of the head in fused passes (model.py:376-377 + 388-389): BatchNorm2d (batch statistics) + ReLU applied inside
    the Dropout + Conv2d(k=1) kernels, forward and backward, against float64 autograd; with dropout on, against the separate kernels
    (same counter-hash mask)."""
    HW, M = H * W, B * H * W
    x, xr = bt(rnd(M, C, seed=51) * 1.5 + 0.3, split)
    g, b = 1 + 0.1 * rnd(C, seed=52), 0.1 * rnd(C, seed=53)
    w, cb = rnd(ncls, C, seed=54, scale=C**-0.5), rnd(ncls, seed=55)
    rm0, rv0 = 0.1 * rnd(C, seed=56), 1 + 0.2 * torch.rand(C)
    rm, rv = rm0.clone().to(DEV), rv0.clone().to(DEV)
    scale, shift, mean, rstd = (torch.empty(C, device=DEV) for _ in range(4))
    sums = torch.empty(2 * C, dtype=torch.float64, device=DEV)
    logits = torch.empty(B, ncls, H, W, device=DEV)
    # flat "gradient buffer": dw | db | dgamma | dbeta (registered in the deterministic mode: fixed-point shadow + ordered partial folds)
    nw = ncls * C
    flat = torch.zeros(nw + ncls + 2 * C + 8, device=DEV)
    dw, db, dgam, dbet = flat[:nw].view(ncls, C), flat[nw : nw + ncls], flat[nw + ncls : nw + ncls + C], flat[nw + ncls + C : nw + ncls + 2 * C]
    xd = xr.clone().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    wd, cbd = w.double().requires_grad_(True), cb.double().requires_grad_(True)
    rmd, rvd = rm0.double().clone(), rv0.double().clone()
    act = F.relu(F.batch_norm(xd, rmd, rvd, gd, bd, True, 0.1, 1e-5))
    ref = (act.view(B, HW, C) @ wd.t() + cbd).permute(0, 2, 1).reshape(B, ncls, H, W)
    dl = rnd(B, ncls, H, W, seed=57)
    gx, gg, gb, gw, gcb = torch.autograd.grad((ref * dl.double()).sum(), [xd, gd, bd, wd, cbd])
    try:
        if det:
            ops.set_deterministic(flat)
        ops.bn_stats(x, g.to(DEV), b.to(DEV), rm, rv, scale, shift, mean, rstd, sums, M, C, True)
        close(rm, rmd, 1e-5, what="running mean")
        close(rv, rvd, 1e-5, what="running var")
        ops.classifier_bn_fwd(x, scale, shift, w.to(DEV), cb.to(DEV), logits, B, HW, C, ncls)
        close(logits, ref.detach(), 3e-5, what="fused logits")
        dx = BT.empty((M, C), split, DEV)
        ops.classifier_bn_bwd(dl.to(DEV), x, scale, shift, mean, rstd, w.to(DEV), dx, dw, db, dgam, dbet, sums, None, B, HW, C, ncls)
        ops.det_fold(0, flat.numel())
        close(dx.float(), gx, tol_out(split), what="fused dx")
        close(dw, gw, 3e-5, what="fused dw")
        close(db, gcb, 3e-5, what="fused db")
        close(dgam, gg, 3e-5, what="fused dgamma")
        close(dbet, gb, 3e-5, what="fused dbeta")
        if det:  # a second run gives the same bits
            first = (dx.float().clone(), flat.clone())
            flat.zero_()
            ops.classifier_bn_bwd(dl.to(DEV), x, scale, shift, mean, rstd, w.to(DEV), dx, dw, db, dgam, dbet, sums, None, B, HW, C, ncls)
            ops.det_fold(0, flat.numel())
            assert torch.equal(dx.float(), first[0]) and torch.equal(flat, first[1])
        # dropout on: the separate kernels (BatchNorm apply -> classifier) regenerate the same mask from the same seed
        y = BT.empty((M, C), split, DEV)
        ops.bn_relu_fwd(x, g.to(DEV), b.to(DEV), rm, rv, y, scale, shift, mean, rstd, sums, M, C, True, False)
        lg_sep, lg_fus = torch.empty_like(logits), torch.empty_like(logits)
        ops.classifier_fwd(y, w.to(DEV), cb.to(DEV), lg_sep, B, HW, C, ncls, seed=77, p=0.1)
        ops.classifier_bn_fwd(x, scale, shift, w.to(DEV), cb.to(DEV), lg_fus, B, HW, C, ncls, seed=77, p=0.1)
        assert (lg_fus - logits).abs().max().item() > 1e-3  # the mask does something
        close(lg_fus, lg_sep.double().cpu(), 3e-5 if split else 4e-3, what="fused vs separate logits with dropout")
        flat2 = torch.zeros_like(flat)
        dw2, db2, dgam2, dbet2 = flat2[:nw].view(ncls, C), flat2[nw : nw + ncls], flat2[nw + ncls : nw + ncls + C], flat2[nw + ncls + C : nw + ncls + 2 * C]
        if det:
            ops.set_deterministic(None)
        df, dx2, dx3 = BT.empty((M, C), split, DEV), BT.empty((M, C), split, DEV), BT.empty((M, C), split, DEV)
        ops.classifier_bwd(dl.to(DEV), y, w.to(DEV), df, dw2, db2, None, B, HW, C, ncls, seed=77, p=0.1)
        ops.bn_relu_bwd(x, df, scale, shift, mean, rstd, dx2, dgam2, dbet2, sums, M, C)
        flat.zero_()
        ops.classifier_bn_bwd(dl.to(DEV), x, scale, shift, mean, rstd, w.to(DEV), dx3, dw, db, dgam, dbet, sums, None, B, HW, C, ncls, seed=77, p=0.1)
        t = 3e-5 if split else 1.2e-2
        close(dx3.float(), dx2.float().double().cpu(), t, what="fused vs separate dx with dropout")
        close(dw, dw2.double().cpu(), t, what="fused vs separate dw with dropout")
        close(db, db2.double().cpu(), 3e-5, what="fused vs separate db with dropout")
        close(dgam, dgam2.double().cpu(), t, what="fused vs separate dgamma with dropout")
        close(dbet, dbet2.double().cpu(), t, what="fused vs separate dbeta with dropout")
    finally:
        ops.set_deterministic(None)


def test_ce_loss_all_ignored_and_float_labels():
    B, ncls, H, W = 1, 2, 8, 8
    logits = rnd(B, ncls, H, W, seed=42).to(DEV)
    stats = torch.zeros(2, dtype=torch.float64, device=DEV)
    lab = torch.full((B, H, W), -1.0, device=DEV)  # reference labels are float tensors (.long() at segmentation.py:116)
    dlog = torch.ones_like(logits)
    ops.ce_loss(logits, lab, None, -1, stats, dlog)
    assert stats[1].item() == 0 and stats[0].item() == 0 and dlog.abs().max().item() == 0


def test_adamw_matches_torch():
    from oracle import prithvi_oracle as O

    n = 4096 + 8
    p0, g = rnd(n, seed=43), rnd(n, seed=44)
    p_ref = p0.clone()
    m_ref, v_ref = torch.zeros(n), torch.zeros(n)
    p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    hyper = torch.zeros(16, device=DEV)
    hyper[:5] = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 1e-2])
    hyper[11], hyper[12] = 1 - 0.9, 1 - 0.999
    shadow = BT.empty((n,), True, DEV)
    for step in range(1, 4):
        O.adamw_step(p_ref, g, m_ref, v_ref, step, lr=1e-3, wd=1e-2)
        ops.adamw_advance(hyper)
        ops.adamw_step(p, g.to(DEV), m, v, shadow, hyper, n)
    close(p, p_ref, 1e-6, what="adamw p")
    close(m, m_ref, 1e-6, what="adamw m")
    close(v, v_ref, 1e-6, what="adamw v")
    close(shadow.float(), p_ref, 2e-5, what="adamw shadow")
    # reference torch.optim.AdamW
    q = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([q], lr=1e-3, weight_decay=1e-2)
    for _ in range(3):
        q.grad = g.clone()
        opt.step()
    close(p, q.detach(), 1e-6, what="adamw vs torch.optim")
    # weight clipping (base.py:103-113)
    hyper[7], hyper[8], hyper[9] = -0.5, 0.5, 1.0
    ops.adamw_advance(hyper)
    ops.adamw_step(p, g.to(DEV), m, v, None, hyper, n)
    assert p.min().item() >= -0.5 and p.max().item() <= 0.5


@pytest.mark.parametrize("T", [1, 3])
@pytest.mark.parametrize("dtype", [torch.int16, torch.float32])
def test_normalize_chips(T, dtype):
    from oracle import prithvi_oracle as O

    B, C, H, W = 2, 6, 32, 24
    g = torch.Generator().manual_seed(7)
    raw = torch.randint(0, 10000, (B, T * C, H, W), generator=g).to(dtype)
    mean = [0.14, 0.13, 0.12, 0.31, 0.20, 0.12]
    std = [0.04, 0.04, 0.05, 0.08, 0.07, 0.05]
    out = ops.normalize_chips(raw.to(DEV), torch.tensor(mean, device=DEV), torch.tensor(std, device=DEV), T, constant_multiplier=1e-4)
    for b in range(B):
        chip = raw[b].numpy().astype(np.float64) * 1e-4  # dataloader.py:737
        ref = O.normalize_chip(chip, mean, std, T)
        assert np.allclose(out[b].cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    assert out.shape == (B, C, T, H, W)


def test_bad_arguments_raise():
    from instageo_amd._lib import HipLibraryError

    x = BT.empty((8, 12), False, DEV)
    with pytest.raises(HipLibraryError):
        ops.linear_fwd(x, x, None, x, 8, 12, 12)  # K not a multiple of 8
    with pytest.raises(HipLibraryError):
        ops.attention_fwd(x, x, None, 1, 8, 1, hd=32)
    with pytest.raises(HipLibraryError):
        ops.linear_fwd(BT.empty((8, 16), True, DEV), BT.empty((8, 16), False, DEV), None, BT.empty((8, 8), False, DEV), 8, 8, 16)


def test_auc_histograms_and_softmax_prob():
    """ig_auc_update / ig_softmax_prob against the oracle (and through it the reference RunningAUC fixture)."""
    import os

    from instageo_amd.metrics import RunningAUC, auc_from_histograms
    from oracle import prithvi_oracle as O

    for ncls, nbins in [(2, 1024), (13, 1024), (3, 64)]:  # 13 x 1024 exceeds the LDS histogram budget -> global atomics
        B, H, W = 3, 40, 56
        g = torch.Generator().manual_seed(31 + ncls)
        logits = torch.randn(B, ncls, H, W, generator=g) * 2
        labels = torch.randint(-1, ncls, (B, H, W), generator=g)
        auc = RunningAUC(ncls, n_bins=nbins, ignore_index=-1, device=DEV)
        auc.update_from_logits(logits.to(DEV), labels.to(DEV))
        auc.update_from_logits(logits.to(DEV), labels.float().to(DEV))  # float labels as the reference datasets deliver them
        probs = torch.softmax(logits, 1).permute(0, 2, 3, 1).reshape(-1, ncls).numpy()
        y = labels.reshape(-1).numpy()
        keep = y != -1
        pos, neg = O.auc_histograms(y[keep], probs[keep], ncls, nbins)
        # softmax differs from torch's in the last ulp: a score within 1e-6 of a bin edge may land in the neighbour bin
        assert np.abs(auc.pos_hist - 2 * pos).sum() <= 8 and np.abs(auc.neg_hist - 2 * neg).sum() <= 8 * ncls
        assert auc.pos_hist.sum() == 2 * pos.sum() and auc.neg_hist.sum() == 2 * neg.sum()
        ref_macro, ref_per = O.auc_score(pos, neg)
        got = auc.score()
        assert abs(got["roc_auc_macro"] - ref_macro) < 1e-4  # tolerance: bin-edge flips only
        assert np.allclose(auc_from_histograms(pos, neg), ref_per, rtol=0, atol=1e-15, equal_nan=True)
        auc.reset()
        assert auc.pos_hist.sum() == 0
    # host-array path with the reference signature reproduces the reference fixture exactly
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "auc.npz"))
    auc = RunningAUC(3, device=DEV)
    auc.update(z["y_true"][:1500], z["probs"][:1500])
    auc.update(z["y_true"][1500:], z["probs"][1500:])
    assert np.array_equal(auc.pos_hist, z["pos_hist"]) and np.array_equal(auc.neg_hist, z["neg_hist"])
    sc = auc.score()
    assert sc["roc_auc_macro"] == float(z["macro"]) and sc["roc_auc_per_class"] == z["per_class"].tolist()
    with pytest.raises(ValueError):
        RunningAUC(3, device=DEV).update(np.zeros(4), np.zeros(4))
    # predict_step's probability map
    logits = torch.randn(2, 2, 32, 48, generator=torch.Generator().manual_seed(2))
    close(ops.softmax_prob(logits.to(DEV), 1), torch.softmax(logits.double(), 1)[:, 1], 1e-6, what="softmax prob")


@pytest.mark.parametrize("use_log", [False, True])
def test_mse_loss_and_regression_metrics(use_log):
    """ig_mse_loss (masked MSE + gradient + streaming metric sums) against the oracle restatement of regression.py /
    RunningRegressionMetrics; the host-array path against the reference fixture."""
    import os

    from instageo_amd.metrics import RunningRegressionMetrics
    from oracle import prithvi_oracle as O

    B, H, W = 3, 64, 80
    g = torch.Generator().manual_seed(41)
    out = torch.rand(B, 1, H, W, generator=g) * 1.5
    lab = torch.rand(B, H, W, generator=g) * 2.0
    lab[torch.rand(B, H, W, generator=g) < 0.1] = -100.0
    stats = torch.zeros(2, dtype=torch.float64, device=DEV)
    dl = torch.empty(B, 1, H, W, device=DEV)
    met = RunningRegressionMetrics(include_ee=True, device=DEV)
    ops.mse_loss(out.to(DEV), lab.to(DEV), -100.0, use_log, stats, dl, met.device_sums(DEV), met.ee_bias, met.ee_coef, True)
    od = out.double().clone().requires_grad_(True)
    loss, preds, l2 = O.regression_loss(od, lab.double(), -100.0, use_log)
    loss.backward()
    n = int((lab != -100.0).sum())
    assert int(stats[1].item()) == n
    assert abs(stats[0].item() / n - loss.item()) <= 2e-6 * max(1.0, loss.item())  # fp32 arithmetic per element, fp64 sums
    close(dl / n, od.grad, 2e-6, what="mse grad")
    ref = O.regression_metrics(O.regression_sums(l2.numpy(), preds.numpy()), include_ee=True)
    got = met.compute()
    for k in ("mae", "rmse", "r2_score", "pearson_corrcoef"):
        assert abs(got[k] - ref[k]) <= 2e-5 * max(1.0, abs(ref[k])), k  # expm1f/log1pf of the device vs torch in float64
    assert abs(got["ee_percentage"] - ref["ee_percentage"]) <= 0.05  # boundary pixels of |e| <= bias + coef x
    if not use_log:
        z = np.load(os.path.join(os.path.dirname(__file__), "golden", "regression.npz"))
        m2 = RunningRegressionMetrics(include_ee=True, device=DEV)
        m2.update(z["y_true"][:1200], z["y_pred"][:1200])
        m2.update(z["y_true"][1200:], z["y_pred"][1200:])
        c = m2.compute()
        got2 = np.array([c[k] for k in ("mae", "rmse", "r2_score", "pearson_corrcoef", "ee_percentage")])
        assert np.allclose(got2, z["metrics"], rtol=5e-6, atol=0.05)  # the device path holds the values in float32
        assert m2.n == 3000
        m2.reset()
        assert m2.n == 0 and np.isnan(m2.compute()["mae"])


def test_wgrad8_plan_cache_is_pointer_free_and_capture_safe():
    """gemm8w.hip keys its plans by SHAPE (the tables hold offsets, the operand pointers travel as kernel arguments): 2000 forward +
    backward passes of ``torch.ops.instageo_mi355x.linear`` on freshly allocated tensors leave the free device memory where it was
    (round 3 leaked two device tables per pointer set), and a shape the engine has not planned yet, met inside a stream capture,
    falls back without allocating (the capture stays valid and the replay computes the right gradient)."""
    from instageo_amd import torch_ops

    torch_ops.register()
    ns = torch.ops.instageo_mi355x
    M, N, K = 512, 256, 256
    w = (rnd(N, K, seed=2) * K**-0.5).to(DEV).bfloat16().requires_grad_(True)
    b = rnd(N, seed=3).to(DEV).requires_grad_(True)

    def one():
        x = torch.randn(M, K, device=DEV).bfloat16().requires_grad_(True)  # fresh operand / gradient buffers every pass
        y, _ = ns.linear(x, w, b, 0)
        gx, gw, gb = torch.autograd.grad(y, (x, w, b), torch.randn(M, N, device=DEV).bfloat16())
        return gw

    for _ in range(20):
        one()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    keep = []
    for i in range(2000):
        g = one()
        if i % 250 == 0:
            keep.append(g)  # hold some results so the caching allocator really hands out new addresses
    torch.cuda.synchronize()
    del keep, g
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, f"device memory shrank by {(free0 - free1) / 2**20:.1f} MiB over 2000 passes"

    # more distinct shapes than the cache holds (96): least-recently-used plans are evicted (their tables freed), results stay right,
    # and an evicted shape is simply planned again
    xe, xer = bt(rnd(256 + 110 * 64, 256, seed=8), False)
    dye, dyer = bt(rnd(256 + 110 * 64, 256, seed=9), False)
    dwe = torch.zeros(256, 256, device=DEV)
    for rep in range(2):
        for j in range(0, 110, 1 if rep == 0 else 37):
            Me = 256 + 64 * j
            dwe.zero_()
            ops.linear_wgrad(BT(dye.hi[:Me], None), BT(xe.hi[:Me], None), dwe, Me, 256, 256)
            if j % 27 == 0:
                assert ops.last_kernel().startswith(("gemm8w_kernel", "gemm4w_kernel")), ops.last_kernel()
                close(dwe, dyer[:Me].t() @ xer[:Me], 3e-5, what=f"linear wgrad M={Me} (plan cache eviction)")
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] > free1 - (96 << 20)  # (the slab workspace of the widest token split: grown geometrically, kept)

    # a cold shape inside a capture: the grouped engine must not allocate or copy synchronously there
    M2, N2, K2 = 1408, 512, 768  # not used anywhere else in the suite
    dy, dyr = bt(rnd(M2, N2, seed=5), False)
    x2, xr2 = bt(rnd(M2, K2, seed=6), False)
    dw = torch.zeros(N2, K2, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            ops.linear_wgrad(dy, x2, dw, M2, N2, K2)
    torch.cuda.current_stream().wait_stream(side)
    dw.zero_()
    graph.replay()
    torch.cuda.synchronize()
    close(dw, dyr.t() @ xr2, 3e-5, what="linear wgrad captured on a cold shape")


def test_torch_library_functional_ops_autograd():
    """``torch.ops.instageo_mi355x.linear`` / ``layer_norm`` (torch_ops.py): dispatcher ops with register_autograd; forward and
    gradients against float64 torch on the bf16-rounded operands; torch.library.opcheck validates schema / fake / autograd
    registration.  The raw mutating ops are exercised through the same calls."""
    from instageo_amd import torch_ops

    torch_ops.register()
    ns = torch.ops.instageo_mi355x
    M, N, K = 300, 256, 192
    x = rnd(M, K, seed=31).to(DEV).bfloat16().requires_grad_(True)
    w = (rnd(N, K, seed=32) * K**-0.5).to(DEV).bfloat16().requires_grad_(True)
    b = rnd(N, seed=33).to(DEV).requires_grad_(True)
    for act in (0, 1):
        y, _ = ns.linear(x, w, b, act)
        xd, wd, bd = (t.detach().double().cpu().requires_grad_(True) for t in (x, w, b))
        pre = xd @ wd.t() + bd
        ref = F.gelu(pre) if act else pre
        close(y.float(), ref.detach(), tol_out(False), what=f"ops.linear act={act}")
        dy = rnd(M, N, seed=34).to(DEV).bfloat16()
        gx, gw, gb = torch.autograd.grad(y, (x, w, b), dy)
        rx, rw, rb = torch.autograd.grad(ref, (xd, wd, bd), dy.double().cpu())
        close(gx.float(), rx, 2e-2, what="ops.linear dx")  # dy * gelu' is rounded to bf16 before the two gradient GEMMs
        close(gw.float(), rw, 2e-2, what="ops.linear dw")
        close(gb, rb, 2e-2, what="ops.linear dbias")
    xs = (rnd(197, 256, seed=35) * 2 + 0.5).to(DEV).requires_grad_(True)
    g = (1 + 0.1 * rnd(256, seed=36)).to(DEV).requires_grad_(True)
    be = (0.1 * rnd(256, seed=37)).to(DEV).requires_grad_(True)
    out, mean, rstd = ns.layer_norm(xs, g, be, 1e-5)
    xd, gd, bd = (t.detach().double().cpu().requires_grad_(True) for t in (xs, g, be))
    ref = F.layer_norm(xd, (256,), gd, bd, 1e-5)
    close(out.float(), ref.detach(), tol_out(False), what="ops.layer_norm")
    dy = rnd(197, 256, seed=38).to(DEV).bfloat16()
    gx, gg, gb = torch.autograd.grad(out, (xs, g, be), dy)
    rx, rg, rb = torch.autograd.grad(ref, (xd, gd, bd), dy.double().cpu())
    close(gx, rx, 2e-5, what="ops.layer_norm dx")
    close(gg, rg, 2e-5, what="ops.layer_norm dgamma")
    close(gb, rb, 2e-5, what="ops.layer_norm dbeta")
    for op, args in ((ns.linear.default, (x.detach(), w.detach(), b.detach(), 1)), (ns.layer_norm.default, (xs.detach(), g.detach(), be.detach(), 1e-5))):
        torch.library.opcheck(op, args, test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    yy = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    torch.library.opcheck(ns.linear_fwd.default, (x.detach(), None, w.detach(), None, b.detach(), yy, None, None, None, M, N, K, 0),
                          test_utils=("test_schema", "test_faketensor"))


def test_loss_statistics_are_bit_reproducible():
    """The grid-wide sums of ig_mse_loss / ig_kd_mse_loss / ig_kd_loss / ig_ce_loss are added in a fixed order (workgroup partials
    folded by the workgroup that arrives last): repeated launches on the same inputs give bit-identical statistics -- many workgroups
    (the 1024-workgroup cap) and a launch with a single one."""
    g = torch.Generator(device=DEV).manual_seed(11)
    for n_img in (40, 1):
        B, H, W, ncls = n_img, 224, 224, 5
        pred = torch.randn(B, 1, H, W, device=DEV, generator=g) * 3
        teach = torch.randn(B, 1, H, W, device=DEV, generator=g) * 3
        lab = torch.rand(B, H, W, device=DEV, generator=g) * 10
        lab[torch.rand(B, H, W, device=DEV, generator=g) < 0.1] = -1.0
        s_log = torch.randn(B, ncls, H, W, device=DEV, generator=g)
        t_log = torch.randn(B, ncls, H, W, device=DEV, generator=g)
        lab_c = torch.randint(-1, ncls, (B, H, W), device=DEV, generator=g)
        runs = []
        for _ in range(4):
            st = torch.zeros(2, dtype=torch.float64, device=DEV)
            ms = torch.zeros(9, dtype=torch.float64, device=DEV)
            kd = torch.zeros(1, dtype=torch.float64, device=DEV)
            kl = torch.zeros(1, dtype=torch.float64, device=DEV)
            ce = torch.zeros(2, dtype=torch.float64, device=DEV)
            ops.mse_loss(pred, lab, -1.0, False, st, None, ms, 0.1, 0.15, True)
            ops.kd_mse_loss(pred, teach, lab, -1.0, False, kd)
            ops.kd_loss(s_log, t_log, lab_c, -1, kl, None)
            ops.ce_loss(s_log, lab_c, None, -1, ce)
            runs.append(torch.cat([st, ms, kd, kl, ce]))
        for r in runs[1:]:
            assert torch.equal(r, runs[0]), (r, runs[0])
        valid = lab != -1.0
        want = ((pred[:, 0] - lab)[valid].double() ** 2).sum().item()
        assert abs(runs[0][0].item() - want) <= 1e-9 * want and runs[0][1].item() == valid.sum().item()
