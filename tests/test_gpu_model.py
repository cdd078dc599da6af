"""End-to-end parity of the HIP path (-m gpu) against the CPU oracle and the committed golden fixtures.

Tolerances (stated per north_star):
* precision "bf16x3" (split-bf16 MFMA): logits within 1e-3 abs of the fp32 reference -- the north-star bar.
* precision "bf16" (the fast/bench mode): bf16 has 8 mantissa bits; through 12+ residual blocks a 1e-3 abs
  bound on O(1) logits is not attainable (a CPU emulation of the same roundings gives ~3e-2 max).  Its
  deviation is *measured* here and bounded by max-abs <= 6e-2, mean-abs <= 8e-3, argmax agreement >= 99 %.
* gradients: the fp32 reference's own autograd noise on these cases is ~3e-3 relative (fixture key
  grad_fp32_noise__*), so bf16x3 gradients are held to 1e-2 relative L2 against the fp64 fixture.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from instageo_amd import ops  # noqa: E402
from instageo_amd.model import PrithviSeg  # noqa: E402
from instageo_amd.segmentation import FusedAdamW, PrithviSegmentationModule, segmentation_loss  # noqa: E402
from oracle import prithvi_oracle as O  # noqa: E402
from oracle.cases import CASES, GRAD_KEYS, case_config, class_weights_for, make_inputs, sub  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def build(name, precision, freeze=False):
    variant, T, ncls, B, depth = CASES[name]
    cfg = case_config(name)
    sd = O.make_state_dict(cfg, seed=1042)
    net = PrithviSeg(temporal_step=T, image_size=224, num_classes=ncls, load_pretrained_weights=False, freeze_backbone=freeze,
                     variant=variant, depth=depth, precision=precision, device=DEV)
    net.load_state_dict(sd, strict=True)
    img, lab = make_inputs(name, cfg, B)
    return cfg, sd, net, img, lab


def report(tag, got, ref):
    d = (got.double().cpu() - ref.double()).abs()
    print(f"[{tag}] max {d.max().item():.3e} mean {d.mean().item():.3e} (ref absmax {ref.abs().max().item():.3f})")
    return d.max().item(), d.mean().item()


@pytest.mark.parametrize("name", ["tiny_t1_c2", "tiny_t3_c13", "v1_100_t1_c2", "v1_100_t3_c13", "v2_300_t1_c2", "v2_600_t1_c2",
                                  "v2_600_full_t1_c2"])  # the last one: prithvi_eo_v2_600 at its full 32 blocks (630M parameters)
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_eval_logits_parity(name, precision):
    cfg, sd, net, img, lab = build(name, precision)
    net.eval()
    with torch.no_grad():
        logits = net(img.to(DEV))
        ref = O.prithvi_seg_forward(cfg, sd, img, training=False)
    mx, mean = report(f"{name}/{precision} eval logits", logits, ref)
    gold = np.load(os.path.join(GOLD, f"{name}.npz"))
    gmx = np.abs(sub(logits) - gold["eval_logits_sub"]).max()
    agree = (logits.argmax(1).cpu() == ref.argmax(1)).float().mean().item()
    print(f"   vs golden fixture {gmx:.3e}; argmax agreement {agree:.5f}")
    if precision == "bf16x3":
        assert mx <= 1e-3, f"bf16x3 logits differ from the fp32 reference by {mx}"
        assert gmx <= 1e-3
        assert agree >= 0.9995
    else:
        # bf16 operands: measured on MI355X over the six cases max 2.5-3.9e-2, mean 3.4-5.4e-3, argmax agreement 0.991-0.998 on
        # random-init logits (std ~0.05: maximally fragile argmax); bounds = 1.5 x the worst measured value
        assert mx <= 6e-2 and mean <= 8e-3 and agree >= 0.99


def test_custom_head_widths_embed_dims():
    """PrithviSeg(embed_dims=[...]) (model.py:304,380-389): a decode head whose widths are not D / 2^i -- eval logits and one
    train-mode backward against the oracle network with the same widths (bf16x3: 1e-3 on logits, 1e-2 rel-L2 on gradients)."""
    import dataclasses

    dims = [256, 96, 64, 40, 24]
    cfg = dataclasses.replace(O.make_config("prithvi_eo_tiny", 1, 2, 224), embed_dims=tuple(dims))
    sd = O.make_state_dict(cfg, seed=7)
    net = PrithviSeg(temporal_step=1, image_size=224, num_classes=2, load_pretrained_weights=False, freeze_backbone=False,
                     variant="prithvi_eo_tiny", embed_dims=dims, precision="bf16x3", device=DEV)
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == O.state_dict_shapes(cfg)
    net.load_state_dict(sd, strict=True)
    img, lab = make_inputs("tiny_t1_c2", cfg, 2)
    net.eval()
    with torch.no_grad():
        mx, _ = report("embed_dims eval logits", net(img.to(DEV)), O.prithvi_seg_forward(cfg, sd, img, training=False))
    assert mx <= 1e-3
    net.cfg.drop_p = 0.0
    net.train()
    eng = net.engine
    logits = eng.forward(img.to(DEV), training=True, save=True)
    stats = torch.zeros(2, dtype=torch.float64, device=DEV)
    dlog = torch.empty_like(logits)
    cw = class_weights_for(2).to(DEV)
    ops.ce_loss(logits, lab.to(DEV), cw, -1, stats, dlog)
    net.store.ensure_grad().zero_()
    eng.backward(dlog, count=stats)
    sd64 = {k: (v.double().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k and not k.endswith("pos_embed") else v.double() if v.dtype.is_floating_point else v)
            for k, v in sd.items()}
    loss = O.seg_loss(O.prithvi_seg_forward(cfg, sd64, img.double(), training=True), lab, cw.cpu().double(), -1)
    keys = ["segmentation_head.0.0.weight", "segmentation_head.1.2.weight", "segmentation_head.3.0.bias", "segmentation_head.5.weight",
            "prithvi_encoder.blocks.0.attn.qkv.weight"]
    grads = torch.autograd.grad(loss, [sd64[k] for k in keys])
    for k, g in zip(keys, grads):
        got = net.store.entries[k].api_view(net.store.grad).double().cpu()
        err = ((got - g).norm() / g.norm()).item()
        print(f"   embed_dims grad {k:44s} rel-L2 {err:.3e}")
        assert err <= 1e-2, k
    with pytest.raises(ValueError):
        PrithviSeg(variant="prithvi_eo_tiny", load_pretrained_weights=False, embed_dims=[128, 96, 64, 40, 24], device=DEV)  # dims[0] != D * T


@pytest.mark.parametrize("B", [108, 216, 432])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_benchmark_batch_ties_to_the_golden_fixture_and_is_permutation_equivariant(precision, B):
    """BASELINE configs[1] at its benchmark batches (432 chips = bench.py's default, 216 / 108 = the defaults of earlier rounds: the sizes at
    which the 8-phase GEMM engine, the persistent tile walks -- one and two rounds of 256 CUs -- and the full-occupancy attention
    grids run).  Eval mode has no cross-sample coupling, so
    (i) the logits of the fixture's four chips, placed at scattered batch positions, must equal the reference-generated golden
        vector (tests/golden/v1_100_t1_c2.npz, written from the imported reference) -- 1e-3 in bf16x3, the bf16 bound otherwise;
    (ii) permuting the batch permutes the logits BIT FOR BIT (no kernel's arithmetic depends on where a chip sits)."""
    name = "v1_100_t1_c2"
    cfg, sd, net, img4, _ = build(name, precision)
    net.eval()
    g = torch.Generator().manual_seed(5)
    img = torch.randn((B, *img4.shape[1:]), generator=g)
    pos = [0, 37, 71, B - 1]
    img[pos] = img4
    with torch.no_grad():
        logits = net(img.to(DEV))
        perm = torch.randperm(B, generator=g)
        logits_p = net(img[perm].to(DEV))
    assert torch.equal(logits_p.cpu(), logits.cpu()[perm]), "eval logits depend on the batch position of a chip"
    gold = np.load(os.path.join(GOLD, f"{name}.npz"))
    got4 = logits[pos].cpu()
    gmx = np.abs(sub(got4) - gold["eval_logits_sub"]).max()
    with torch.no_grad():
        l4 = net(img4.to(DEV)).cpu()  # the same chips as a batch of 4 (other engines: below the 8-phase tile threshold)
    d = (got4 - l4).abs().max().item()
    print(f"[{precision}] B={B} vs golden {gmx:.3e}; vs the B=4 engines {d:.3e}")
    if precision == "bf16x3":
        assert gmx <= 1e-3 and d <= 2e-4
    else:
        assert gmx <= 8e-2 and d <= 8e-2


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_300m_benchmark_batch_ties_to_the_golden_fixture(precision):
    """BASELINE configs[4] (Prithvi-V2-300M: D = 1024, 24 blocks, K = 1024 / 4096 linears) at bench.py's per-GPU batch of 32 chips
    (6304 tokens: 300 / 400 / 100 tiles of 256 x 256).  The B = 1 fixture runs the 128 x 128 instances only; here the fixture's chip sits at
    scattered positions of a random batch (eval mode has no cross-sample coupling):
    (i) its logits equal the reference-generated golden vector (tests/golden/v2_300_t1_c2.npz) within 1e-3 in bf16x3 (the bf16 bound otherwise),
    (ii) a batch permutation permutes the logits bit for bit, and
    (iii) the launch log shows the 256-wide GEMM instances (gemm8.hip's 256 x 256 kernel -- 6 template fields in its name -- or gemm4.hip) and
          the 8-phase convolution engine at D = 1024 / K = 4096."""
    B = 32
    name = "v2_300_t1_c2"
    cfg, sd, net, img1, _ = build(name, precision)
    net.eval()
    g = torch.Generator().manual_seed(17)
    img = torch.randn((B, *img1.shape[1:]), generator=g)
    pos = [0, 13, B - 1]
    for q in pos:
        img[q] = img1[0]
    ops.profile_begin(["ig_conv3x3_fwd", "ig_convT_fwd", "ig_linear_fwd", "ig_linear_residual_fwd", "ig_attention_fwd"])
    with torch.no_grad():
        logits = net(img.to(DEV))
    names = sorted(ops.profile_end()["kernels"])
    print("   kernels:", names)
    wide = [k for k in names if k.startswith("gemm4_kernel") or (k.startswith("gemm8_kernel") and k.count(",") == 5)]
    assert wide, f"no 256-wide GEMM instance ran at B = {B}: {names}"
    if precision == "bf16":
        assert any(k.startswith("gemm4_kernel") for k in names), names  # plain bf16 store / residual kinds with >= 128 tiles
    assert any(k.startswith(("conv4_kernel", "conv8_kernel")) for k in names), f"the wide-convolution engines did not run at B = {B}: {names}"
    with torch.no_grad():
        perm = torch.randperm(B, generator=g)
        logits_p = net(img[perm].to(DEV))
    assert torch.equal(logits_p.cpu(), logits.cpu()[perm]), "eval logits depend on the batch position of a chip"
    gold = np.load(os.path.join(GOLD, f"{name}.npz"))
    for q in pos:
        gmx = np.abs(sub(logits[q:q + 1].cpu()) - gold["eval_logits_sub"]).max()
        print(f"   [{precision}] B={B} chip at {q}: vs golden {gmx:.3e}")
        assert gmx <= (1e-3 if precision == "bf16x3" else 8e-2)
    assert torch.equal(logits[pos[0]].cpu(), logits[pos[1]].cpu()) and torch.equal(logits[pos[0]].cpu(), logits[pos[2]].cpu())


@pytest.mark.parametrize("B", [36, 72])
def test_multitemporal_benchmark_batch_ties_to_the_golden_fixture(B):
    """BASELINE configs[2] (T = 3, 13 classes: configs/multitemporal_crop_classification.yaml:14-30) at the batches bench.py times it
    at (36 / 72 chips per GPU).  At one chip the head's row counts (196 ... 50 176) keep most stages off the wide-convolution
    engines (conv8.hip, gemm8w.hip modes 1 / 2), so the B = 1 fixture does not exercise the benchmarked path.  Here the fixture's
    chip sits at scattered positions of a random batch; eval mode has no cross-sample coupling, so
    (i) its logits equal the reference-generated golden vector (tests/golden/v1_100_t3_c13.npz) within 1e-3 (bf16x3),
    (ii) a batch permutation permutes the logits bit for bit, and
    (iii) the launch log shows that the wide-convolution engines (conv4 / conv8) served the head (kernel names as rocprofv3 prints them)."""
    name = "v1_100_t3_c13"
    cfg, sd, net, img1, _ = build(name, "bf16x3")
    net.eval()
    g = torch.Generator().manual_seed(11)
    img = torch.randn((B, *img1.shape[1:]), generator=g)
    pos = [0, 17, B - 1]
    for q in pos:
        img[q] = img1[0]
    ops.profile_begin(["ig_conv3x3_fwd", "ig_convT_fwd", "ig_linear_fwd", "ig_attention_fwd"])
    with torch.no_grad():
        logits = net(img.to(DEV))
    kern = ops.profile_end()["kernels"]
    names = sorted(kern)
    print("   kernels:", names)
    assert any(k.startswith(("conv4_kernel", "conv8_kernel")) for k in names), f"the wide-convolution engines did not run at B = {B}: {names}"
    assert any(k.startswith("gemm8_kernel") for k in names) and any(k.startswith("attn2_") for k in names)
    with torch.no_grad():
        perm = torch.randperm(B, generator=g)
        logits_p = net(img[perm].to(DEV))
    assert torch.equal(logits_p.cpu(), logits.cpu()[perm]), "eval logits depend on the batch position of a chip"
    gold = np.load(os.path.join(GOLD, f"{name}.npz"))
    for q in pos:
        gmx = np.abs(sub(logits[q:q + 1].cpu()) - gold["eval_logits_sub"]).max()
        print(f"   B={B} chip at {q}: vs golden {gmx:.3e}")
        assert gmx <= 1e-3
    assert torch.equal(logits[pos[0]].cpu(), logits[pos[1]].cpu()) and torch.equal(logits[pos[0]].cpu(), logits[pos[2]].cpu())


@pytest.mark.parametrize("name", ["tiny_t1_c2", "tiny_t3_c13"])
def test_stage_activations_bf16x3(name):
    """Per-stage check (features image layout c = d*T+t, head stages) against the golden sub-samples."""
    cfg, sd, net, img, lab = build(name, "bf16x3")
    net.eval()
    gold = np.load(os.path.join(GOLD, f"{name}.npz"))
    with torch.no_grad():
        logits, feats = net(img.to(DEV), return_features=True)
    assert feats.shape == (img.shape[0], cfg.embed_dim * cfg.num_frames, 14, 14)
    assert np.abs(sub(feats) - gold["features_sub"]).max() <= 1e-3
    ws = net.engine._last["ws"]
    head0 = ws["f"][1].float().permute(0, 3, 1, 2).contiguous()
    head3 = ws["f"][4].float().permute(0, 3, 1, 2).contiguous()
    assert np.abs(sub(head0) - gold["stage_head0_sub"]).max() <= 1e-3
    assert np.abs(sub(head3) - gold["stage_head3_sub"]).max() <= 1e-3
    x_enc = ws["x_in"][1].view(img.shape[0], cfg.tokens, cfg.embed_dim)
    assert np.abs(sub(x_enc) - gold["stage_block0_sub"]).max() <= 1e-3


def rel_l2(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-300)).item()


@pytest.mark.parametrize("name", ["tiny_t1_c2", "tiny_t3_c13", "v1_100_t1_c2", "v1_100_t3_c13", "v2_300_t1_c2", "v2_600_t1_c2",
                                  "v1_100_t1_c2_b16", "v1_100_t3_c13_b8"])  # the last two: the reference YAMLs' own batch sizes
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_train_step_gradients(name, precision):
    """forward(train-mode BN, dropout p=0) + loss + backward: loss, logits and gradients vs the fp64 fixture that
    oracle/gen_golden.py computed with the REFERENCE network (tiny, the model-size Prithvi-100M cases -- D = 768 runs the
    8-phase / ping-pong / dual-K GEMM engines end to end against reference-generated gradients -- Prithvi-V2-300M:
    D = 1024, 24 blocks, 16 heads, the 64-channel head of BASELINE configs[4] -- and the 600M shape family at depth 2:
    D = 1280, 16 heads of 80, patch 14 / 257 tokens, head kernels [5, 5, 5, 7], model.py:154-177)."""
    cfg, sd, net, img, lab = build(name, precision)
    net.cfg.drop_p = 0.0
    net.train()
    gold = np.load(os.path.join(GOLD, f"{name}.npz"))
    cw = class_weights_for(cfg.num_classes).to(DEV)
    eng = net.engine
    rm_before = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k}
    logits = eng.forward(img.to(DEV), training=True, save=True)
    stats = torch.zeros(2, dtype=torch.float64, device=DEV)
    dlog = torch.empty_like(logits)
    conf = torch.zeros(cfg.num_classes, cfg.num_classes, dtype=torch.int64, device=DEV)
    ops.ce_loss(logits, lab.to(DEV), cw, -1, stats, dlog, None, None, conf)
    net.store.ensure_grad().zero_()
    eng.backward(dlog, count=stats)
    loss = (stats[0] / stats[1]).item()
    lerr = np.abs(sub(logits) - gold["train_logits_sub"]).max()
    print(f"[{name}/{precision}] train logits err {lerr:.3e}; loss {loss:.6f} vs {float(gold['train_loss']):.6f}")
    worst = 0.0
    for k in GRAD_KEYS:
        g = net.store.entries[k].api_view(net.store.grad)
        got = sub(g.contiguous(), 1024)
        ref = gold["grad_sub__" + k]
        err = np.linalg.norm(got.astype(np.float64) - ref) / (np.linalg.norm(ref) + 1e-300)
        nrm = g.double().norm().item() / float(gold["grad_norm__" + k])
        worst = max(worst, err)
        print(f"   grad {k:48s} rel-L2 {err:.3e}  norm ratio {nrm:.5f}  (fp32 ref noise {float(gold['grad_fp32_noise__' + k]):.1e})")
        tol = 1e-2 if precision == "bf16x3" else 0.25  # measured worst key: 4.3-8.4e-3 (bf16x3), 0.145-0.165 (bf16: 1.5 x headroom)
        assert err <= tol, f"{k}: rel L2 {err}"
    # mIoU / confusion (metrics.py semantics) from the device histogram
    from instageo_amd.metrics import metrics_from_matrix

    m = metrics_from_matrix(conf.cpu().numpy())
    print(f"   mIoU {m['jaccard']:.6f} vs {float(gold['miou']):.6f}; acc {m['accuracy']:.6f} vs {float(gold['acc']):.6f}")
    if precision == "bf16x3":
        assert lerr <= 1e-3 and abs(loss - float(gold["train_loss"])) <= 1e-3
        assert abs(m["jaccard"] - float(gold["miou"])) <= 1e-3 and abs(m["accuracy"] - float(gold["acc"])) <= 1e-3
        assert int(conf.sum().item()) == int(gold["confusion"].sum())
    else:
        # bf16: train-mode BatchNorm amplifies the operand rounding on the logits (measured 4.3-6.6e-2), but loss and mIoU -- the
        # quantities north_star bounds at 1e-3 -- stay within 3.4e-4 / 2.5e-4 of the fp64 reference on all six cases
        assert lerr <= 9e-2 and abs(loss - float(gold["train_loss"])) <= 1e-3
        assert abs(m["jaccard"] - float(gold["miou"])) <= 1e-3
    # BatchNorm running statistics (momentum 0.1, unbiased variance) vs the oracle
    upd = {}
    with torch.no_grad():
        O.prithvi_seg_forward(cfg, sd, img, training=True, bn_momentum_update=upd)
    new_sd = net.state_dict()
    for k, v in upd.items():
        assert (new_sd["" + k].cpu() - v).abs().max().item() <= (1e-3 if precision == "bf16x3" else 5e-2), k
        assert not torch.equal(new_sd[k].cpu(), rm_before[k].cpu())
    assert int(new_sd["segmentation_head.0.3.num_batches_tracked"].item()) == 1


def test_autograd_path_matches_fused():
    """`loss.backward()` through PrithviSeg.forward (compatible path) gives the same grads as the engine."""
    name = "tiny_t1_c2"
    cfg, sd, net, img, lab = build(name, "bf16x3")
    net.cfg.drop_p = 0.0
    net.train()
    cw = class_weights_for(cfg.num_classes).to(DEV)
    logits = net(img.to(DEV))
    assert logits.requires_grad
    loss = segmentation_loss(logits, lab.to(DEV), cw, -1)
    loss.backward()
    gold = np.load(os.path.join(GOLD, f"{name}.npz"))
    assert abs(loss.item() - float(gold["train_loss"])) <= 1e-3
    named = dict(net.named_parameters())
    for k in GRAD_KEYS:
        got = sub(named[k].grad.contiguous(), 1024)
        ref = gold["grad_sub__" + k]
        err = np.linalg.norm(got.astype(np.float64) - ref) / (np.linalg.norm(ref) + 1e-300)
        assert err <= 1e-2, (k, err)


def test_whole_network_custom_op_opcheck_and_torch_compile():
    """``PrithviSeg.forward`` dispatches through ONE custom op, ``torch.ops.instageo_mi355x.prithvi_seg`` (north_star: "behind
    PyTorch-ROCm custom ops"; the reference's seam is the module, base.py:28,69-77): ``torch.library.opcheck`` validates its schema,
    fake implementation and autograd registration; ``torch.compile(net, fullgraph=True)`` captures the network as one node (no graph
    break on the ctypes layer) and its eval logits equal eager BIT FOR BIT; ``loss.backward()`` through the op fills ``p.grad``."""
    from instageo_amd import torch_ops

    name = "tiny_t1_c2"
    cfg, sd, net, img, lab = build(name, "bf16x3")
    torch_ops.register()
    op = torch.ops.instageo_mi355x.prithvi_seg
    x = img.to(DEV)
    net.eval()
    params = [p.detach() for _, p in net._flat_params()]
    torch.library.opcheck(op.default, (x, params, net._handle, False, False, True),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    with torch.no_grad():
        eager = net(x)
        eager_f = net(x, return_features=True)[1]
        ref = O.prithvi_seg_forward(cfg, sd, img, training=False)
    assert (eager.cpu() - ref).abs().max().item() <= 1e-3
    import torch._dynamo as dynamo

    dynamo.reset()
    for backend in ("inductor", "aot_eager"):
        try:
            comp = torch.compile(net, fullgraph=True, backend=backend)
            with torch.no_grad():
                got = comp(x)
            break
        except Exception as e:  # an image without a working inductor toolchain: the graph capture is what is under test
            if backend == "aot_eager":
                raise
            print(f"   inductor unavailable here ({type(e).__name__}); falling back to aot_eager")
            dynamo.reset()
    assert torch.equal(got, eager), "compiled eval logits differ from eager"
    explain = dynamo.explain(net)(x)
    assert explain.graph_break_count == 0 and explain.graph_count == 1, str(explain)
    ops_seen = [n.target for g in explain.graphs for n in g.graph.nodes if n.op == "call_function"]
    assert any("prithvi_seg" in str(t) for t in ops_seen), ops_seen
    assert eager_f.shape == (img.shape[0], cfg.embed_dim * cfg.num_frames, 14, 14)
    # autograd through the op == the engine's own backward
    net.cfg.drop_p = 0.0
    net.train()
    for p in net.parameters():
        p.grad = None
    cw = class_weights_for(cfg.num_classes).to(DEV)
    out = net(x)
    assert out.grad_fn is not None
    loss = segmentation_loss(out, lab.to(DEV), cw, -1)
    loss.backward()
    gold = np.load(os.path.join(GOLD, f"{name}.npz"))
    for k in ("segmentation_head.5.weight", "prithvi_encoder.blocks.0.attn.qkv.weight"):
        g = dict(net.named_parameters())[k].grad
        err = np.linalg.norm(sub(g.contiguous(), 1024).astype(np.float64) - gold["grad_sub__" + k]) / np.linalg.norm(gold["grad_sub__" + k])
        print(f"   grad through the op {k}: rel-L2 {err:.3e}")
        assert err <= 1e-2, k
    eager_grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    # a loss on the features output must not train nothing silently: the op refuses a gradient into it
    out, feats = net(x, return_features=True)
    with pytest.raises(RuntimeError, match="not differentiable"):
        (out.mean() + feats.mean()).backward()
    # COMPILED training steps: the forward generation is an op output, so the backward graph traced once stays valid for every later
    # step (read from Python state in setup_context it was baked in at trace time and the first real step raised)
    dynamo.reset()
    comp = torch.compile(net, fullgraph=True, backend="aot_eager")
    for step in range(3):
        for p in net.parameters():
            p.grad = None
        loss = segmentation_loss(comp(x), lab.to(DEV), cw, -1)
        loss.backward()
        for k, p in net.named_parameters():
            assert torch.equal(p.grad, eager_grads[k]), f"compiled step {step}: gradient of {k} differs from eager"


def test_frozen_backbone_and_state_dict_roundtrip(tmp_path):
    name = "tiny_t1_c2"
    cfg, sd, net, img, lab = build(name, "bf16", freeze=True)
    assert all(not p.requires_grad for p in net.prithvi_encoder.parameters())
    assert all(p.requires_grad for p in net.segmentation_head.parameters())
    mod_sd = net.state_dict()
    assert list(mod_sd.keys()) == list(O.state_dict_shapes(cfg).keys())
    for k, v in sd.items():
        assert torch.equal(mod_sd[k].cpu(), v), k
    torch.save({"state_dict": {k: v.cpu().contiguous() for k, v in mod_sd.items()}}, tmp_path / "c.ckpt")
    net2 = PrithviSeg(temporal_step=1, num_classes=2, load_pretrained_weights=False, freeze_backbone=True, variant="prithvi_eo_tiny", device=DEV)
    net2.load_state_dict(torch.load(tmp_path / "c.ckpt")["state_dict"], strict=True)
    net.eval(), net2.eval()
    with torch.no_grad():
        assert torch.equal(net(img.to(DEV)), net2(img.to(DEV)))
    # frozen backbone: a fused step must leave encoder weights untouched and move head weights
    mod = PrithviSegmentationModule(freeze_backbone=True, load_pretrained_weights=False, num_classes=2, model_name="prithvi_eo_tiny",
                                    class_weights=[1, 3], ignore_index=-1, device=DEV)
    mod.net.load_state_dict(sd)
    before = mod.net.store.flat.clone()
    mod.fused_train_step(img.to(DEV), lab.to(DEV))
    after = mod.net.store.flat
    ee = mod.net.store.encoder_end
    assert torch.equal(before[:ee], after[:ee]) and not torch.equal(before[ee:], after[ee:])


def test_three_fused_steps_track_reference_training():
    """3 x (forward, CE, backward, AdamW) vs the fp32 CPU restatement with torch.optim.AdamW (dropout p=0)."""
    name = "tiny_t1_c2"
    cfg = case_config(name)
    sd = O.make_state_dict(cfg, seed=1042)
    img, lab = make_inputs(name, cfg, 2)
    mod = PrithviSegmentationModule(freeze_backbone=False, load_pretrained_weights=False, num_classes=2, model_name="prithvi_eo_tiny",
                                    class_weights=[1, 3], ignore_index=-1, learning_rate=1e-3, precision="bf16x3", device=DEV)
    mod.net.load_state_dict(sd)
    mod.net.cfg.drop_p = 0.0
    # reference loop
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k and not k.endswith("pos_embed")]
    ref_p = {k: sd[k].clone().requires_grad_(True) for k in names}
    opt = torch.optim.AdamW(list(ref_p.values()), lr=1e-3, weight_decay=1e-2)
    cw = torch.tensor([1.0, 3.0])
    full = dict(sd)
    ref_losses, losses = [], []
    for _ in range(3):
        full.update(ref_p)
        upd = {}
        out = O.prithvi_seg_forward(cfg, full, img, training=True, bn_momentum_update=upd)
        loss = O.seg_loss(out, lab, cw, -1)
        opt.zero_grad()
        loss.backward()
        opt.step()
        full.update(upd)
        ref_losses.append(loss.item())
        st = mod.fused_train_step(img.to(DEV), lab.to(DEV))
        losses.append((st[0] / st[1]).item())
    print("losses", losses, "reference", ref_losses)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 5e-3
    assert losses[-1] < losses[0]
    new = mod.net.state_dict()
    # first AdamW steps move every weight by ~lr regardless of gradient scale: compare the *updates*
    for k in ["segmentation_head.5.weight", "segmentation_head.3.2.weight", "prithvi_encoder.blocks.0.attn.qkv.weight"]:
        d_ref = ref_p[k].detach() - sd[k]
        d_got = new[k].cpu() - sd[k]
        cos = torch.nn.functional.cosine_similarity(d_ref.reshape(1, -1), d_got.reshape(1, -1)).item()
        print(f"   update cosine {k}: {cos:.5f}")
        assert cos >= 0.98, (k, cos)


def test_module_api_and_metrics_names():
    name = "tiny_t1_c2"
    cfg = case_config(name)
    img, lab = make_inputs(name, cfg, 2)
    mod = PrithviSegmentationModule(freeze_backbone=False, load_pretrained_weights=False, num_classes=2, model_name="prithvi_eo_tiny",
                                    class_weights=[1, 3], ignore_index=-1, device=DEV)
    (opt,), scheds = mod.configure_optimizers()
    assert isinstance(opt, FusedAdamW) and isinstance(scheds[0], torch.optim.lr_scheduler.CosineAnnealingWarmRestarts)
    batch = (img.to(DEV), lab.to(DEV).float())  # reference labels are float tensors
    loss = mod.training_step(batch, 0)
    assert loss.requires_grad and torch.isfinite(loss)
    opt.zero_grad()
    loss.backward()
    opt.step()
    vloss = mod.validation_step(batch, 0)
    tloss = mod.test_step(batch, 0)
    assert not vloss.requires_grad and not tloss.requires_grad
    probs = mod.predict_step(batch[0])
    assert probs.shape == (2, 224, 224) and probs.min() >= 0 and probs.max() <= 1
    mod.on_train_epoch_end(), mod.on_validation_epoch_end(), mod.on_test_epoch_end()
    for st in ("train", "val", "test"):
        for m in ("loss", "Acc", "IoU", "F1", "Precision", "Recall", "IoU_0", "IoU_1", "F1_0", "F1_1"):
            assert f"{st}_{m}" in mod.logged, f"{st}_{m}"
    assert mod.train_metrics.total == 0  # reset at epoch end
    ck = mod.checkpoint_state_dict()
    assert "criterion.weight" in ck and "net.prithvi_encoder.cls_token" in ck and "net.segmentation_head.5.bias" in ck


def test_input_validation():
    net = PrithviSeg(temporal_step=1, num_classes=2, load_pretrained_weights=False, freeze_backbone=True, variant="prithvi_eo_tiny", device=DEV)
    net.eval()
    with torch.no_grad():
        a = net(torch.zeros(1, 6, 224, 224, device=DEV))  # 4-D input accepted when T == 1 (pritvhi.py:507-509)
        assert a.shape == (1, 2, 224, 224)
        assert net(torch.zeros(0, 6, 1, 224, 224, device=DEV)).shape == (0, 2, 224, 224)  # empty batch
        with pytest.raises(ValueError):
            net(torch.zeros(1, 5, 1, 224, 224, device=DEV))
        with pytest.raises(Exception):
            net(torch.zeros(1, 6, 1, 224, 224))  # CPU tensor: no CPU fallback
    with pytest.raises(KeyError):
        PrithviSeg(variant="prithvi_eo_v3_900", load_pretrained_weights=False, device=DEV)
    with pytest.raises(RuntimeError):
        PrithviSeg(variant="prithvi_eo_tiny", load_pretrained_weights=True, device=DEV)


@pytest.mark.parametrize("variant,freeze,steps,B", [("prithvi_eo_tiny", False, 60, 8), ("prithvi_eo_v1_100", True, 240, 4)])
def test_trained_weights_bf16_miou_and_loss_vs_oracle(variant, freeze, steps, B):
    """north_star: "outputs (logits, mIoU) must match the reference CPU path within 1e-3".  Random-init logits have std ~0.05,
    so their argmax is maximally fragile; this test first TRAINS for 60 fused steps (bf16 kernels, lr 1e-3) on a learnable
    synthetic task (label = sign of a low-frequency field in band 0) -- Prithvi-tiny end to end, and Prithvi-100M with a frozen
    random backbone (the reference's default freeze_backbone=True: only the decode head trains) -- then evaluates the trained
    weights on held-out chips three ways: CPU oracle fp32, the bf16x3 parity mode and the bf16 bench mode.  bf16x3 is held to the
    1e-3 bar on logits, loss and mIoU in both cases; bf16 meets the loss / mIoU bar on the tiny model and misses it on the 100M case
    (asserted at twice the measured deviation)."""
    from instageo_amd.metrics import metrics_from_matrix

    cfg = O.make_config(variant, 1, 2)
    cw = class_weights_for(2)

    def batch(seed, B=B):
        g = torch.Generator().manual_seed(seed)
        blocks = torch.randn(B, 6, 1, 14, 14, generator=g)
        x = blocks.repeat_interleave(16, 3).repeat_interleave(16, 4) + 0.3 * torch.randn(B, 6, 1, 224, 224, generator=g)
        y = (blocks[:, 0, 0] > 0).long().repeat_interleave(16, 1).repeat_interleave(16, 2)
        y[torch.rand(B, 224, 224, generator=g) < 0.05] = -1
        return x, y

    mod = PrithviSegmentationModule(freeze_backbone=freeze, load_pretrained_weights=False, num_classes=2, model_name=variant,
                                    class_weights=[1, 3], ignore_index=-1, learning_rate=2e-3 if freeze else 1e-3, scheduler=False,
                                    precision="bf16", device=DEV)
    mod.net.load_state_dict(O.make_state_dict(cfg, seed=1042))
    first = last = None
    for step in range(steps):
        x, y = batch(100 + step % 12)
        st = mod.fused_train_step(x.to(DEV), y.to(DEV))
        loss = (st[0] / st[1]).item()
        first = loss if first is None else first
        last = loss
    assert last < (0.75 if freeze else 0.6) * first, f"the task did not train: loss {first:.4f} -> {last:.4f}"
    sd = {k: v.detach().cpu().clone() for k, v in mod.net.state_dict().items()}
    xv, yv = batch(999)
    with torch.no_grad():
        ref = O.prithvi_seg_forward(cfg, sd, xv, training=False)
    ref_loss = O.seg_loss(ref, yv, cw, -1).item()
    ref_m = O.confusion_metrics(O.confusion_matrix(yv.numpy(), ref.argmax(1).numpy(), 2, -1))
    conf_ref = ref.softmax(1).max(1).values.mean().item()
    print(f"[trained {variant}] train loss {first:.4f} -> {last:.4f}; oracle eval loss {ref_loss:.5f} mIoU {ref_m['jaccard']:.5f} acc {ref_m['accuracy']:.5f} mean max-prob {conf_ref:.3f}")
    # a trained, confident model (random init: ~0.33).  The frozen-backbone case over-fits its 48 training chips (held-out loss ~1.8 at 0.98
    # mean confidence) and its held-out mIoU moves with rounding-level changes of the training trajectory (same build, same box: 0.505 with every fusion switched off, 0.565 /
    # 0.484 with the BatchNorm statistics taken by the separate pass / from the convolution's epilogue -- identical sums up to order): the bar only
    # makes sure the weights are far from initialisation; the assertions that matter are the mode-vs-oracle differences below
    assert ref_m["jaccard"] > (0.42 if freeze else 0.6)
    out = {}
    for precision in ("bf16x3", "bf16"):
        net = PrithviSeg(temporal_step=1, num_classes=2, load_pretrained_weights=False, freeze_backbone=freeze, variant=variant,
                         precision=precision, device=DEV)
        net.load_state_dict(sd)
        net.eval()
        with torch.no_grad():
            logits = net(xv.to(DEV))
        conf = torch.zeros(2, 2, dtype=torch.int64, device=DEV)
        loss = segmentation_loss(logits, yv.to(DEV), cw.to(DEV), -1, confusion=conf).item()
        m = metrics_from_matrix(conf.cpu().numpy())
        dl, dm, dx = abs(loss - ref_loss), abs(m["jaccard"] - ref_m["jaccard"]), (logits.cpu() - ref).abs().max().item()
        agree = (logits.argmax(1).cpu() == ref.argmax(1)).float().mean().item()
        print(f"   {precision:7s}: |dloss| {dl:.2e}  |dmIoU| {dm:.2e}  max|dlogits| {dx:.2e}  argmax agreement {agree:.6f}")
        out[precision] = (dl, dm, dx, agree)
        del net
    dl, dm, dx, agree = out["bf16x3"]
    assert dl <= 1e-3 and dm <= 1e-3 and dx <= 1e-3, "bf16x3 misses the 1e-3 bar on trained weights"
    dl, dm, dx, agree = out["bf16"]
    if not freeze:  # Prithvi-tiny trained end to end: bf16 also meets the bar on loss and mIoU (measured 2e-5 / 2e-5)
        assert dl <= 1e-3 and dm <= 1e-3 and agree >= 0.999, f"bf16 on trained weights: dloss {dl} dmIoU {dm} agreement {agree}"
    else:
        # Prithvi-100M, random frozen backbone + trained head: the head reads O(1) features through 12 un-trained blocks and the
        # bf16 operand rounding reaches the loss (measured |dloss| 2.0e-2, |dmIoU| 6.3e-3, argmax agreement 0.9959): on this case
        # the 1e-3 bar is met by bf16x3 ONLY.  Bounds = 2 x measured.
        assert dl <= 4e-2 and dm <= 1.3e-2 and agree >= 0.99, f"bf16 on trained weights: dloss {dl} dmIoU {dm} agreement {agree}"


def test_tl_variant_matches_the_oracle_and_its_scales_stay_put():
    """prithvi_eo_v2_300_tl (2 blocks for speed): logits == oracle (pinned against the reference, whose forward ignores the coordinate
    encoders); a fused train step leaves the two gradient-less ``scale`` parameters untouched, as torch.optim.AdamW does."""
    cfg = O.make_config("prithvi_eo_v2_300_tl", 1, 2, 224, 2)
    sd = O.make_state_dict(cfg, seed=1042)
    mod = PrithviSegmentationModule(freeze_backbone=False, load_pretrained_weights=False, num_classes=2, model_name="prithvi_eo_v2_300_tl",
                                    depth=2, class_weights=[1, 3], ignore_index=-1, precision="bf16x3", device=DEV)
    mod.net.load_state_dict(sd, strict=True)
    img, lab = make_inputs("tiny_t1_c2", cfg, 2)
    mod.net.eval()
    with torch.no_grad():
        out = mod.net(img.to(DEV))
        ref = O.prithvi_seg_forward(cfg, sd, img, training=False)
    assert (out.cpu() - ref).abs().max().item() <= 1e-3
    before = {k: v.clone() for k, v in mod.net.state_dict().items() if k.endswith("_embed_enc.scale")}
    w0 = mod.net.state_dict()["prithvi_encoder.blocks.0.attn.qkv.weight"].clone()
    mod.fused_train_step(img.to(DEV), lab.to(DEV))
    after = mod.net.state_dict()
    assert all(torch.equal(after[k], v) for k, v in before.items()) and len(before) == 2
    assert not torch.equal(after["prithvi_encoder.blocks.0.attn.qkv.weight"], w0)


def test_other_input_sizes_interpolate_the_position_table():
    """The reference accepts any square input that is a multiple of the patch size: ``interpolate_pos_encoding`` resamples the
    position table (bicubic, align_corners=True; pritvhi.py:149-203) and PrithviSeg.forward reshapes by the actual token count
    (model.py:406-413).  Eval logits and gradients at 160 and 256 pixels against the oracle, which calls the same torch routine."""
    cfg, sd, net, img, lab = build("tiny_t1_c2", "bf16x3")
    net.eval()
    for S in (160, 256):
        g = torch.Generator().manual_seed(S)
        x = torch.randn(2, 6, 1, S, S, generator=g)
        with torch.no_grad():
            out = net(x.to(DEV))
            ref = O.prithvi_seg_forward(O.make_config("prithvi_eo_tiny", 1, 2, 224), sd, x, training=False)
        assert out.shape == (2, 2, S, S) and ref.shape == out.shape
        assert (out.cpu() - ref).abs().max().item() <= 1e-3, f"size {S}"
    net.cfg.drop_p = 0.0
    net.train()
    x = torch.randn(2, 6, 1, 160, 160, generator=torch.Generator().manual_seed(1))
    y = torch.randint(0, 2, (2, 160, 160), generator=torch.Generator().manual_seed(2))
    loss = segmentation_loss(net(x.to(DEV)), y.to(DEV), class_weights_for(2).to(DEV), -1)
    loss.backward()
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    _, ref_loss, ref_g = O.train_step_reference(O.make_config("prithvi_eo_tiny", 1, 2, 224), sd64, x.double(), y, class_weights_for(2).double(), -1,
                                                ["prithvi_encoder.blocks.0.attn.qkv.weight", "segmentation_head.3.2.weight"])
    assert abs(loss.item() - ref_loss.item()) <= 1e-3
    got = dict(net.named_parameters())
    for k, rg in ref_g.items():
        assert rel_l2(got[k].grad, rg) <= 1e-2, k
    with pytest.raises(ValueError):
        net(torch.zeros(1, 6, 1, 100, 100, device=DEV))  # not a multiple of the patch size


def _train_steps(name, precision, deterministic, B, steps=3):
    variant, T, ncls, _, depth = CASES[name]
    cfg = case_config(name)
    sd = O.make_state_dict(cfg, seed=1042)
    img, lab = make_inputs(name, cfg, B)
    mod = PrithviSegmentationModule(freeze_backbone=False, load_pretrained_weights=False, num_classes=ncls, model_name=variant,
                                    temporal_step=T, depth=depth, class_weights=class_weights_for(ncls).tolist(), ignore_index=-1,
                                    learning_rate=1e-3, precision=precision, device=DEV)
    mod.net.load_state_dict(sd)
    mod.net.engine.deterministic = deterministic
    stats = []
    for _ in range(steps):  # dropout stays ON: its mask is a hash of (seed, step counter, element), not a random stream
        st = mod.fused_train_step(img.to(DEV), lab.to(DEV))
        stats.append(st.clone())
    torch.cuda.synchronize()
    return mod.net.store.flat.clone(), mod.net.store.grad.clone(), torch.stack(stats)


@pytest.mark.parametrize("name,precision,B", [("tiny_t3_c13", "bf16x3", 3), ("v1_100_t1_c2", "bf16", 6), ("v1_100_t1_c2", "bf16x3", 4),
                                              ("v1_100_t3_c13", "bf16", 2), ("v2_600_t1_c2", "bf16", 2)])
def test_deterministic_mode_is_bit_reproducible(name, precision, B):
    """engine.deterministic (the reference's Trainer(deterministic=True), pipeline_utils.py:373): two runs of three full training
    steps from the same weights and batch end in bit-identical parameters, gradients and loss statistics.  Every
    multi-contributor float reduction of the step (bias / norm / head gradients, BatchNorm and loss sums) takes an
    order-independent route in this mode; the weight gradients of the linears are ordered folds in both modes."""
    try:
        p1, g1, s1 = _train_steps(name, precision, True, B)
        p2, g2, s2 = _train_steps(name, precision, True, B)
        assert ops.deterministic()
        assert torch.equal(s1, s2), (s1, s2)
        assert torch.equal(g1, g2), f"{(g1 != g2).sum().item()} gradient elements differ"
        assert torch.equal(p1, p2), f"{(p1 != p2).sum().item()} parameters differ"
        # and it is the same training: the gradients of ONE step agree with the default (float-atomic) mode to summation-order
        # rounding (later steps are not comparable element by element: AdamW's first updates are +-lr whatever the gradient's size,
        # so a last-bit difference in a near-zero gradient moves that weight by 2 lr -- the default mode does that to itself)
        _, g1s, s1s = _train_steps(name, precision, True, B, steps=1)
        ops.set_deterministic(None)  # the registration is per process: it stays with the last deterministic engine until cleared
        assert not ops.deterministic()
        _, g0s, s0s = _train_steps(name, precision, False, B, steps=1)
        rel = ((g1s.double() - g0s.double()).norm() / g0s.double().norm()).item()
        dl = ((s1s[:, 0] / s1s[:, 1]) - (s0s[:, 0] / s0s[:, 1])).abs().max().item()
        _, g0, _ = _train_steps(name, precision, False, B)
        _, g0b, _ = _train_steps(name, precision, False, B)
        print(f"[{name}/{precision}] deterministic vs default, one step: grad rel-L2 {rel:.3e}, |dloss| {dl:.3e}; default mode, two runs "
              f"of three steps: {(g0 != g0b).sum().item()} of {g0.numel()} gradient elements differ, rel-L2 "
              f"{((g0.double() - g0b.double()).norm() / g0.double().norm()).item():.3e}")
        assert rel <= 1e-5 and dl <= 1e-6
    finally:
        ops.set_deterministic(None)


def test_det_fold_adds_fixed_point_shadow():
    """ig_set_deterministic / ig_det_fold: colsum into a registered buffer lands in the int64 shadow (2^44 fixed point), the fold
    moves it into the buffer and clears the shadow; a target outside the buffer keeps the float atomic."""
    torch.manual_seed(0)
    M, C = 5000, 96
    x = torch.randn(M, C, device=DEV)
    xb = ops.BT.from_float(x, False)
    ref = xb.float().double().sum(0).float()
    buf = torch.zeros(4 * C, device=DEV)
    other = torch.zeros(C, device=DEV)
    try:
        ops.set_deterministic(buf)
        ops.colsum(xb, buf[C : 2 * C], M, C)
        ops.colsum(xb, other, M, C)
        torch.cuda.synchronize()
        assert buf.abs().max().item() == 0.0  # still in the shadow
        sh = ops._DET["shadow"]
        assert (sh[C : 2 * C] != 0).all() and (sh[:C] == 0).all() and (sh[2 * C :] == 0).all()
        got1 = (sh[C : 2 * C].double() * 2.0 ** -44).float()
        ops.det_fold(C, 2 * C)
        torch.cuda.synchronize()
        assert (sh == 0).all()
        assert torch.equal(buf[C : 2 * C], got1) and buf[:C].abs().max().item() == 0.0
        assert torch.allclose(buf[C : 2 * C], ref, rtol=1e-5, atol=1e-4) and torch.allclose(other, ref, rtol=1e-5, atol=1e-4)
        # two shadow runs give the same bits
        ops.colsum(xb, buf[2 * C : 3 * C], M, C)
        ops.det_fold(2 * C, 3 * C)
        assert torch.equal(buf[2 * C : 3 * C], buf[C : 2 * C])
    finally:
        ops.set_deterministic(None)
    assert not ops.deterministic()


def test_divergence_stays_visible_in_deterministic_mode():
    """A diverged step must report NaN / Inf, as the reference's autograd does (Lightning logs a NaN loss).  The deterministic mode sums
    through fixed-point integers, and a float -> integer conversion turns NaN into 0 and saturates Inf: non-finite and
    out-of-range contributions therefore bypass the integer path -- the column sums into a registered gradient buffer, the loss
    statistics of ig_ce_loss, and the classifier bias gradient."""
    torch.manual_seed(0)
    M, C = 3000, 64
    x = torch.randn(M, C, device=DEV)
    x[17, 5] = float("nan")
    x[99, 9] = float("inf")
    x[123, 11] = 3.0e38  # finite but far outside the 2^44 fixed-point range
    xb = ops.BT.from_float(x, False)
    buf = torch.zeros(2 * C, device=DEV)
    try:
        ops.set_deterministic(buf)
        ops.colsum(xb, buf[:C], M, C)
        ops.det_fold(0, C)
        torch.cuda.synchronize()
        assert torch.isnan(buf[5]) and torch.isinf(buf[9]) and buf[9] > 0 and buf[11] > 1e38
        ok = torch.ones(C, dtype=torch.bool)
        ok[[5, 9, 11]] = False
        ref = xb.float().double().sum(0).float()
        assert torch.allclose(buf[:C][ok.to(DEV)], ref[ok.to(DEV)], rtol=1e-5, atol=1e-4)
    finally:
        ops.set_deterministic(None)
    # loss statistics: NaN logits -> NaN loss (and NaN dlogits), both label dtypes' kernel
    B, ncls, HW = 2, 3, 4096
    logits = torch.randn(B, ncls, HW, device=DEV)
    labels = torch.randint(0, ncls, (B, HW), device=DEV)
    cw = torch.ones(ncls, device=DEV)
    for bad in (float("nan"), float("inf")):
        lg = logits.clone()
        lg[1, 2, 77] = bad
        stats = torch.zeros(2, dtype=torch.float64, device=DEV)
        dl = torch.empty_like(lg)
        ops.ce_loss(lg, labels, cw, -1, stats, dl)
        torch.cuda.synchronize()
        assert not torch.isfinite(stats[0]), f"loss sum {stats[0].item()} for a {bad} logit"
        assert stats[1].item() == B * HW
        assert not torch.isfinite(dl[1, :, 77]).all()
        stats.zero_()
        ops.ce_loss(logits, labels, cw, -1, stats, dl)  # the poison word was re-armed: the next launch is clean
        assert torch.isfinite(stats[0])


@pytest.mark.parametrize("name,precision,B", [("tiny_t1_c2", "bf16", 4), ("v1_100_t1_c2", "bf16", 6)])
def test_graphed_train_step_equals_eager_bit_for_bit(name, precision, B):
    """make_graphed_train_step (one hipGraph replay per step; warm-up and capture leave no trace in the training state) against
    the eager fused step: with the deterministic reductions three replays and three eager steps from the same weights end in
    bit-identical parameters and loss statistics -- the capture holds the fixed-point shadow folds, the range tables and the
    fresh-step zeroing like any other launch."""
    variant, T, ncls, _, depth = CASES[name]
    cfg = case_config(name)
    sd = O.make_state_dict(cfg, seed=1042)
    img, lab = make_inputs(name, cfg, B)
    x, y = img.to(DEV), lab.to(DEV)

    def module():
        mod = PrithviSegmentationModule(freeze_backbone=False, load_pretrained_weights=False, num_classes=ncls, model_name=variant,
                                        temporal_step=T, depth=depth, class_weights=class_weights_for(ncls).tolist(), ignore_index=-1,
                                        learning_rate=1e-3, precision=precision, device=DEV)
        mod.net.load_state_dict(sd)
        mod.net.engine.deterministic = True
        return mod

    try:
        eager = module()
        s_eager = torch.stack([eager.fused_train_step(x, y).clone() for _ in range(3)])
        p_eager = eager.net.store.flat.clone()
        graphed = module()
        run = graphed.make_graphed_train_step(x, y)
        s_graph = torch.stack([run(x, y).clone() for _ in range(3)])
        torch.cuda.synchronize()
        assert torch.equal(s_graph, s_eager), (s_graph, s_eager)
        assert torch.equal(graphed.net.store.flat, p_eager), f"{(graphed.net.store.flat != p_eager).sum().item()} parameters differ"
    finally:
        ops.set_deterministic(None)
