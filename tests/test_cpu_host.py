"""CPU tests (-m "not gpu") of the host side: C-ABI library loads and exports every declared symbol, the
PrithviSeg state_dict / checkpoint contract, config surface, and that nothing computes without the GPU."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "instageo-e2e-geospatial-ml_amd")


@pytest.fixture(scope="session")
def built_lib():
    from instageo_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-j4"], check=True)
    return _lib


def test_library_exports_every_declared_symbol(built_lib):
    lib = built_lib.load()
    names = built_lib.declared_symbols()
    assert len(names) >= 30
    header = open(built_lib.HEADER_PATH).read()
    for n in re.findall(r"\b(ig_\w+)\s*\(", header):
        assert hasattr(lib, n), n
    assert lib.ig_version() >= 100
    assert built_lib.last_error() == "" or isinstance(built_lib.last_error(), str)


def test_argument_validation_without_gpu(built_lib):
    """IG_REQUIRE rejects bad arguments before any launch, so these calls are safe on a CPU-only box."""
    lib = built_lib.load()
    rc = lib.ig_linear_fwd(None, None, None, None, None, None, None, None, None, 8, 8, 8, 0, None)
    assert rc == -1 and "null pointer" in built_lib.last_error()
    one = ctypes.c_void_p(16)
    rc = lib.ig_linear_fwd(one, None, one, None, None, one, None, None, None, 8, 12, 12, 0, None)
    assert rc == -1 and "multiples of 8" in built_lib.last_error()
    rc = lib.ig_attention_fwd(one, None, one, None, None, 1, 8, 1, 32, None)
    assert rc == -1 and "head_dim" in built_lib.last_error()
    with pytest.raises(built_lib.HipLibraryError):
        built_lib.call("ig_ce_loss", one, one, 7, None, -1, None, None, None, None, None, 1, 4, 99, None)


def test_ops_refuse_cpu_tensors(built_lib):
    from instageo_amd import ops

    x = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(built_lib.HipLibraryError):
        ops.linear_fwd(ops.BT(x), ops.BT(x), None, ops.BT(x), 8, 8, 8)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(PKG, "instageo_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"


@pytest.mark.parametrize("variant,T,ncls", [("prithvi_eo_tiny", 1, 2), ("prithvi_eo_tiny", 3, 13), ("prithvi_eo_v1_100", 1, 2)])
def test_prithviseg_state_dict_contract(variant, T, ncls):
    from instageo_amd.model import PrithviSeg
    from oracle import prithvi_oracle as O

    net = PrithviSeg(temporal_step=T, num_classes=ncls, load_pretrained_weights=False, freeze_backbone=False, variant=variant, device="cpu")
    cfg = O.make_config(variant, T, ncls)
    want = O.state_dict_shapes(cfg)  # verified equal to the reference's keys/shapes by oracle/gen_golden.py
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == want and list(got) == list(want)
    sd = O.make_state_dict(cfg, seed=7)
    net.load_state_dict(sd, strict=True)
    back = net.state_dict()
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    # conv weights are stored [Cout][9][Cin] but exposed with the PyTorch shapes
    e = net.store.entries["segmentation_head.1.2.weight"]
    w = sd["segmentation_head.1.2.weight"]
    assert torch.equal(net.store.flat[e.offset : e.offset + e.numel].view(w.shape[0], 9, w.shape[1]), w.permute(0, 2, 3, 1).reshape(w.shape[0], 9, -1))
    e = net.store.entries["segmentation_head.1.0.weight"]
    w = sd["segmentation_head.1.0.weight"]
    assert torch.equal(net.store.flat[e.offset : e.offset + e.numel].view(w.shape[1], 9, w.shape[0]), w.permute(1, 2, 3, 0).reshape(w.shape[1], 9, -1))
    bad = dict(sd)
    bad.pop("segmentation_head.5.bias")
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad, strict=True)
    assert hasattr(net, "prithvi_encoder") and hasattr(net, "segmentation_head") and net.model_args["num_frames"] == T


def test_tl_variant_carries_the_unused_coordinate_encoder_scales():
    """prithvi_eo_v2_300_tl (model.py:147-153): the reference builds Temporal/LocationEncoder but PrithviViT.forward never calls
    them -- same arithmetic as prithvi_eo_v2_300 plus two (1,) ``scale`` parameters that get no gradient and are never stepped.
    Key order / shapes / zero gradients were checked against the imported reference in the build container;
    the 600M variants (patch 14, head_dim 80, 5x5 / 7x7 head kernels) still raise."""
    from instageo_amd.model import PrithviSeg
    from oracle import prithvi_oracle as O

    net = PrithviSeg(temporal_step=1, num_classes=2, load_pretrained_weights=False, freeze_backbone=False,
                     variant="prithvi_eo_v2_300_tl", depth=1, device="cpu")
    cfg = O.make_config("prithvi_eo_v2_300_tl", 1, 2, 224, 1)
    want = O.state_dict_shapes(cfg)
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == want and list(got) == list(want)
    keys = list(got)
    i = keys.index("prithvi_encoder.patch_embed.proj.bias")
    assert keys[i + 1 : i + 3] == ["prithvi_encoder.temporal_embed_enc.scale", "prithvi_encoder.location_embed_enc.scale"]
    sd = net.state_dict()
    assert float(sd["prithvi_encoder.temporal_embed_enc.scale"]) == pytest.approx(0.1) and net.prithvi_encoder.location_embed_enc.scale.requires_grad
    assert "prithvi_encoder.temporal_embed_enc.scale" not in net.store.entries  # outside the flat buffer: AdamW never touches it
    new = O.make_state_dict(cfg, seed=3)
    net.load_state_dict(new, strict=True)
    assert torch.equal(net.state_dict()["prithvi_encoder.location_embed_enc.scale"], new["prithvi_encoder.location_embed_enc.scale"])
    # the 600M family (model.py:154-177): D = 1280, 16 heads of 80, patch 14, head kernels [5, 5, 5, 7] -- same key / shape contract
    for v in ("prithvi_eo_v2_600", "prithvi_eo_v2_600_tl"):
        n6 = PrithviSeg(load_pretrained_weights=False, variant=v, depth=1, device="cpu")
        c6 = O.make_config(v, 1, 2, 224, 1)
        want6 = O.state_dict_shapes(c6)
        got6 = {k: tuple(t.shape) for k, t in n6.state_dict().items()}
        assert got6 == want6 and list(got6) == list(want6)
        assert got6["segmentation_head.0.2.weight"] == (640, 640, 5, 5) and got6["segmentation_head.3.2.weight"] == (80, 80, 7, 7)
        assert got6["prithvi_encoder.patch_embed.proj.weight"] == (1280, 6, 1, 14, 14) and got6["prithvi_encoder.pos_embed"] == (1, 257, 1280)
        assert n6.cfg.head_sizes == [(16, 32, 30), (30, 60, 58), (58, 116, 114), (114, 228, 224)] and n6.cfg.out_size == 224
        n6.load_state_dict(O.make_state_dict(c6, seed=5), strict=True)


def test_reference_init_statistics():
    from instageo_amd.model import PrithviSeg

    torch.manual_seed(0)
    net = PrithviSeg(temporal_step=1, num_classes=2, load_pretrained_weights=False, freeze_backbone=True, variant="prithvi_eo_tiny", device="cpu")
    sd = net.state_dict()
    assert torch.all(sd["prithvi_encoder.pos_embed"][0, 0] == 0)
    assert torch.all(sd["prithvi_encoder.blocks.0.norm1.weight"] == 1) and torch.all(sd["prithvi_encoder.blocks.0.attn.qkv.bias"] == 0)
    w = sd["prithvi_encoder.blocks.0.mlp.fc1.weight"]
    bound = (6.0 / (w.shape[0] + w.shape[1])) ** 0.5  # xavier_uniform (pritvhi.py:140-143)
    assert w.abs().max() <= bound + 1e-6 and w.abs().max() > 0.9 * bound
    assert abs(sd["prithvi_encoder.cls_token"].std().item() - 0.02) < 0.005
    assert all(not p.requires_grad for p in net.prithvi_encoder.parameters())  # freeze_backbone (model.py:341-343)
    assert all(p.requires_grad for p in net.segmentation_head.parameters())


def test_checkpoint_layout_and_mae_checkpoint_filter(tmp_path):
    from instageo_amd.model import load_prithvi_checkpoint
    from instageo_amd.segmentation import PrithviSegmentationModule

    mod = PrithviSegmentationModule(freeze_backbone=False, load_pretrained_weights=False, num_classes=2, model_name="prithvi_eo_tiny",
                                    class_weights=[1, 3], ignore_index=-1, device="cpu")
    ck = mod.checkpoint_state_dict()
    assert "criterion.weight" in ck and all(k.startswith("net.") or k == "criterion.weight" for k in ck)
    torch.save({"state_dict": ck}, tmp_path / "instageo_best_checkpoint.ckpt")
    mod2 = PrithviSegmentationModule(freeze_backbone=False, load_pretrained_weights=False, num_classes=2, model_name="prithvi_eo_tiny",
                                     class_weights=[1, 3], ignore_index=-1, device="cpu")
    mod2.load_checkpoint_state_dict(torch.load(tmp_path / "instageo_best_checkpoint.ckpt")["state_dict"])
    assert all(torch.equal(a, b) for a, b in zip(mod.net.state_dict().values(), mod2.net.state_dict().values()))
    # a Prithvi MAE checkpoint: encoder.* prefix, decoder keys, mask_token, foreign pos_embed (utils.py:271-315)
    enc = {k[len("prithvi_encoder."):]: v.clone() for k, v in mod.net.state_dict().items() if k.startswith("prithvi_encoder.")}
    mae = {"encoder." + k: (torch.randn_like(v) if v.is_floating_point() else v) for k, v in enc.items()}
    mae["encoder.pos_embed"] = torch.randn(1, 5, 256)
    mae["decoder.blocks.0.norm1.weight"] = torch.zeros(3)
    mae["mask_token"] = torch.zeros(1, 1, 128)
    pos_before = mod2.net.state_dict()["prithvi_encoder.pos_embed"].clone()
    load_prithvi_checkpoint(mod2.net, mae)
    sd2 = mod2.net.state_dict()
    assert torch.equal(sd2["prithvi_encoder.pos_embed"], pos_before)  # model's own fixed table is kept
    assert torch.equal(sd2["prithvi_encoder.blocks.1.mlp.fc1.weight"], mae["encoder.blocks.1.mlp.fc1.weight"])


def test_unsupported_and_pretrained_paths_fail_loudly():
    from instageo_amd.model import PrithviSeg

    with pytest.raises(KeyError):
        PrithviSeg(variant="prithvi_eo_v3_900", load_pretrained_weights=False, device="cpu")
    with pytest.raises(RuntimeError):
        PrithviSeg(variant="prithvi_eo_tiny", load_pretrained_weights=True, device="cpu")
    net = PrithviSeg(variant="prithvi_eo_tiny", load_pretrained_weights=False, device="cpu")
    from instageo_amd._lib import HipLibraryError

    with pytest.raises(HipLibraryError):
        net(torch.zeros(1, 6, 1, 224, 224))
    with pytest.raises(TypeError):
        net.double()


def test_metrics_host_arithmetic_matches_oracle():
    from instageo_amd.metrics import metrics_from_matrix
    from oracle import prithvi_oracle as O

    rng = np.random.default_rng(3)
    cm = rng.integers(0, 50, size=(13, 13))
    cm[5] = 0
    cm[:, 5] = 0  # an absent class -> zero denominators
    a, b = metrics_from_matrix(cm), O.confusion_metrics(cm)
    for k in a:
        assert np.allclose(a[k], b[k], atol=1e-12), k


def test_config_surface():
    from instageo_amd.config import DEFAULTS, check_required_flags, load_config

    # every top-level / nested key of the reference's config.yaml (SURVEY.md 8b "CLI surface to keep")
    assert {"root_dir", "valid_filepath", "train_filepath", "test_filepath", "checkpoint_path", "mode", "is_reg_task", "train", "model",
            "dataloader", "test"} <= set(DEFAULTS)
    assert {"learning_rate", "num_epochs", "batch_size", "class_weights", "ignore_index", "weight_decay", "scheduler", "distillation",
            "teacher_ckpt_path"} <= set(DEFAULTS["train"])
    assert {"bands", "mean", "std", "img_size", "temporal_dim", "replace_label", "reduce_to_zero", "no_data_value", "constant_multiplier",
            "max_pixel_value", "num_workers", "augmentations"} <= set(DEFAULTS["dataloader"])
    assert {"img_size", "crop_size", "stride", "mask_cloud"} <= set(DEFAULTS["test"])
    c = load_config("sen1floods11", ["train.batch_size=4", "mode=eval", "+neptune_experiment_id=abc", "model.weight_clip_range=[-1,1]"])
    assert c["train"]["batch_size"] == 4 and c["train"]["class_weights"] == [1, 3] and c["test"]["img_size"] == 512
    assert c["model"]["model_name"] == "prithvi_eo_v1_100" and c["neptune_experiment_id"] == "abc" and c["model"]["weight_clip_range"] == [-1, 1]
    m = load_config("multitemporal_crop_classification")
    assert m["dataloader"]["temporal_dim"] == 3 and m["model"]["num_classes"] == 13 and len(m["train"]["class_weights"]) == 13
    with pytest.raises(KeyError):
        load_config("config", ["train.nonexistent=1"])
    with pytest.raises(KeyError):
        load_config("nope")
    with pytest.raises(RuntimeError):
        check_required_flags(["root_dir"], {"root_dir": "None"})  # the *string* "None" counts as missing
    check_required_flags(["root_dir"], {"root_dir": "/data"})


def test_torch_library_ops_are_registered_from_the_header():
    """SURVEY.md 8b / north_star: the hot path sits behind PyTorch custom ops.  Every C-ABI entry point that takes device
    pointers is a dispatcher-visible ``instageo_mi355x::`` op whose schema is generated from include/instageo_hip.h
    (non-const pointers = mutated arguments), with a fake implementation; there is no CPU kernel."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode

    from instageo_amd import _lib, torch_ops

    ops_ = torch_ops.register()
    assert torch_ops.register() is ops_  # idempotent
    protos = torch_ops.parse_prototypes()
    tensor_entries = [n for n, ps in protos.items() if n not in torch_ops._SKIP and any("*" in t for t, pn in ps if pn != "stream")]
    assert sorted(ops_) == sorted(n[3:] for n in tensor_entries) and len(ops_) >= 40
    assert set(protos) <= set(_lib.declared_symbols())
    sch = str(torch.ops.instageo_mi355x.linear_fwd.default._schema)
    assert "Tensor? x_hi" in sch and "Tensor(a!)? y_hi" in sch and sch.endswith("int M, int N, int K, int act) -> ()")
    assert "Tensor(a!)? dw" in str(torch.ops.instageo_mi355x.linear_wgrad.default._schema)
    with FakeTensorMode():
        x = torch.empty(64, 32, dtype=torch.bfloat16, device="cuda")
        w = torch.empty(16, 32, dtype=torch.bfloat16, device="cuda")
        y, d = torch.ops.instageo_mi355x.linear(x, w, None, 1)
        assert y.shape == (64, 16) and d.shape == (64, 16) and y.dtype == torch.bfloat16
        assert torch.ops.instageo_mi355x.linear_fwd(x, None, w, None, None, y, None, None, None, 64, 16, 32, 0) is None
        a, m, r = torch.ops.instageo_mi355x.layer_norm(torch.empty(64, 32, device="cuda"), torch.empty(32, device="cuda"), torch.empty(32, device="cuda"), 1e-5)
        assert a.dtype == torch.bfloat16 and m.shape == (64,) and r.shape == (64,)
    import pytest

    with pytest.raises(NotImplementedError):  # no CPU backend
        torch.ops.instageo_mi355x.linear(torch.empty(4, 8, dtype=torch.bfloat16), torch.empty(8, 8, dtype=torch.bfloat16), None, 0)


@pytest.mark.parametrize("variant", ["prithvi_eo_v1_100", "prithvi_eo_v2_300"])
def test_block_gradient_range_tables_partition_the_flat_buffer(variant):
    """Host logic behind the deterministic folds and the fresh-step zeroing (model.py): the "small ranges" of a Block are exactly its
    flat range minus the four weight matrices the grouped weight-gradient kernel writes -- nothing left out, nothing twice -- and the
    notification ranges of a backward pass tile [0, total)."""
    from instageo_amd.model import PrithviSeg

    net = PrithviSeg(variant=variant, load_pretrained_weights=False, device="cpu")
    eng, ent = net.engine, net.engine.store.entries
    L = eng.cfg.depth
    e = "prithvi_encoder."
    starts = [ent[f"{e}blocks.{i}.norm1.weight"].offset for i in range(L)] + [ent[e + "norm.weight"].offset]
    for i in (0, L // 2, L - 1):
        lo, hi = starts[i], starts[i + 1]
        small = eng._block_small_ranges(i, lo, hi)
        big = sorted((ent[f"{e}blocks.{i}.{n}"].offset, ent[f"{e}blocks.{i}.{n}"].offset + ent[f"{e}blocks.{i}.{n}"].numel) for n in eng._BLOCK_WEIGHTS)
        pieces = sorted(small + big)
        assert pieces[0][0] == lo and pieces[-1][1] == hi
        assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:])), "gap or overlap inside a Block"
        assert all(b > a for a, b in small)
        D = eng.cfg.embed_dim
        assert sum(b - a for a, b in small) == 13 * D  # 2 LayerNorms (4 D) + biases of qkv (3 D), proj (D), fc1 (4 D), fc2 (D)
    # the notification ranges of SegEngine.backward: head | final norm | blocks L-1 .. 0 | cls token + patch embedding
    head0 = ent["segmentation_head.0.0.weight"].offset
    tiles = [(head0, eng.store.total), (starts[L], head0)] + [(starts[i], starts[i + 1]) for i in range(L - 1, -1, -1)]
    tiles.append((ent[e + "cls_token"].offset, starts[0]))
    tiles.sort()
    assert tiles[0][0] == 0 and tiles[-1][1] == eng.store.total and all(a[1] == b[0] for a, b in zip(tiles, tiles[1:]))


def test_whole_network_custom_op_is_registered_with_a_fake_implementation():
    """``instageo_mi355x::prithvi_seg`` (torch_ops.py): schema, fake (meta) shapes of logits / features, and no CPU kernel --
    ``PrithviSeg.forward`` dispatches through it (checked on the GPU by tests/test_gpu_model.py)."""
    from torch._subclasses.fake_tensor import FakeTensorMode

    from instageo_amd import torch_ops
    from instageo_amd.model import PrithviSeg

    torch_ops.register()
    op = torch.ops.instageo_mi355x.prithvi_seg
    assert "Tensor[] params" in str(op.default._schema) and "int handle" in str(op.default._schema)
    net = PrithviSeg(variant="prithvi_eo_tiny", temporal_step=3, num_classes=13, load_pretrained_weights=False, device="cpu")
    with FakeTensorMode():
        img = torch.empty(2, 6, 3, 224, 224)
        ps = [torch.empty(p.shape) for _, p in net._flat_params()]
        logits, feats, gen = op(img, ps, net._handle, False, False, True)
        assert logits.shape == (2, 13, 224, 224) and feats.shape == (2, 256 * 3, 14, 14)
        assert gen.shape == (1,) and gen.dtype == torch.int64 and gen.device.type == "cpu"  # the forward generation travels as an op OUTPUT
        grads = torch.ops.instageo_mi355x.prithvi_seg_backward(logits, ps, net._handle, gen)
        assert len(grads) == len(ps)
    with pytest.raises(NotImplementedError):  # no CPU backend
        op(torch.zeros(1, 6, 3, 224, 224), [p.detach() for _, p in net._flat_params()], net._handle, False, False, False)
    handle = net._handle
    del net
    import gc

    gc.collect()
    with pytest.raises(RuntimeError):
        from instageo_amd.model import network_of

        network_of(handle)


def test_m0_is_only_written_by_the_lds_dma_helpers():
    """gemm8.hip / gemm8w.hip / conv8.hip / gemm4.hip set M0 (the LDS destination of an LDS-DMA) without saving or restoring it (the asm statements
    declare the clobber).  That is sound and free only while hipcc keeps nothing of its own in M0 in those kernels: compile each file to gfx950
    assembly (no GPU needed) and check that every M0 reference is one of the helpers' `s_mov_b32 m0, sN` -- or, in the generated K-loops of
    gemm4.hip, gemm8w.hip's gemm4w_kernel and conv8.hip's conv4_kernel, `s_add_u32 m0, sN, imm` -- no read of M0, no other writer."""
    import re
    import shutil
    import subprocess
    import tempfile

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "instageo-e2e-geospatial-ml_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        if not os.path.exists(os.path.join(csrc, "gemm4_gen.inc")):  # normally written by the Makefile
            subprocess.run([sys.executable, os.path.join(csrc, "gen_gemm4.py"), os.path.join(csrc, "gemm4_gen.inc")], check=True)
        for name in ("gemm8", "gemm8w", "conv8", "gemm4"):
            out = os.path.join(tmp, name + ".s")
            subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(root, "include"), "-DIG_HEADER_STAMP=0", "-S",
                            "--cuda-device-only", os.path.join(csrc, name + ".hip"), "-o", out], check=True, capture_output=True)
            refs = [ln.strip() for ln in open(out) if re.search(r"\bm0\b", ln) and not ln.lstrip().startswith(";")]
            assert refs, name
            bad = [ln for ln in refs if not re.fullmatch(r"s_mov_b32 m0, s\d+", ln) and not (name in ("gemm4", "gemm8w", "conv8") and re.fullmatch(r"s_add_u32 m0, s\d+, (0x)?[0-9a-f]+", ln))]
            assert not bad, (name, bad[:5])


def test_gemm4w_generator_emits_a_consistent_instruction_stream(tmp_path):
    """The weight-gradient forms of csrc/gen_gemm4.py (gemm4w_kernel in gemm8w.hip: plain, short-segment and paired blocks).  Structural
    invariants of the generated text:
    * plain segment block: 6 iterations of 128 MFMAs (first / middle / last pair), only the first 64 start from 0; 64 transposed fragment reads
      per iteration (+ 32 at entry); 16 buffer LDS-DMA issues per iteration, each two instructions behind its own M0 write; one vmcnt(8) + one
      barrier per iteration; the descriptors are re-pointed at the next segment exactly once, in front of the last pair; the short block is the
      first-and-last pair alone;
    * paired block: 6 iterations of 192 MFMAs in three products, 64 reads per iteration (+ 48 at entry), 16 DMA issues, and no product's reads
      write a quarter set its own MFMAs consume;
    * every named register lies inside the clobber list; the clobbered VGPR range leaves the low registers to the compiler."""
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "g4.inc"
    subprocess.run([sys.executable, os.path.join(root, "instageo-e2e-geospatial-ml_amd", "csrc", "gen_gemm4.py"), str(out)], check=True)
    text = out.read_text()

    def block(name, end):
        return re.findall(r'"([^"]*?)\\n\\t"', text[text.index("#define " + name):text.index("#define " + end)])

    def check_regs(ins, clob):
        cv = {int(x) for x in re.findall(r'"v(\d+)"', clob)}
        cs = {int(x) for x in re.findall(r'"s(\d+)"', clob)}
        assert {int(x) for x in re.findall(r'"a(\d+)"', clob)} == set(range(256)) and '"m0"' in clob and '"scc"' in clob and '"memory"' in clob
        for i in ins:
            body = re.sub(r"%\[[a-z0-9_]+\]", "", i)
            for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", body):
                assert set(range(int(lo), int(hi) + 1)) <= cv, i
            for r in re.findall(r"\bv(\d+)\b", body):
                assert int(r) in cv, i
            for lo, hi in re.findall(r"\bs\[(\d+):(\d+)\]", body):
                assert set(range(int(lo), int(hi) + 1)) <= cs, i
            for r in re.findall(r"\bs(\d+)\b", body):
                assert int(r) in cs, i
        return cv

    def check_dma(ins, n):
        dma = [k for k, i in enumerate(ins) if i.startswith("buffer_load_dwordx4")]
        assert len(dma) == n
        for k in dma:
            assert ins[k].endswith("offen lds") and ins[k - 2].startswith("s_add_u32 m0, ") and not ins[k - 1].startswith(("s_add_u32 m0", "buffer_load")), ins[k - 2:k + 1]

    MF, RD = "v_mfma_f32_16x16x32_bf16", "ds_read_b64_tr_b16"
    wclob = text[text.index("#define G4W_CLOBBERS"):].split("\n", 1)[0]
    seg, short, pro = block("G4W_ASM_SEG", "G4W_ASM_SEG_SHORT"), block("G4W_ASM_SEG_SHORT", "G4W_CLOBBERS"), block("G4W_ASM_PROLOGUE", "G4W_ASM_SEG")
    for ins, iters in ((seg, 6), (short, 2)):
        mf = [i for i in ins if i.startswith(MF)]
        assert len(mf) == iters * 128 and all(i.endswith(", 0") for i in mf[:64]) and sum(1 for i in mf if i.endswith(", 0")) == 64
        for it in range(iters):
            dst = [int(re.match(MF + r" a\[(\d+):", i).group(1)) for i in mf[it * 128:(it + 1) * 128]]
            assert sorted(dst) == sorted([4 * b for b in range(64)] * 2)
        assert sum(1 for i in ins if i.startswith(RD)) == 32 + iters * 64
        check_dma(ins, iters * 16)
        assert sum(1 for i in ins if i == "s_waitcnt vmcnt(8)") == iters and sum(1 for i in ins if i == "s_barrier") == iters + 1
        assert sum(1 for i in ins if i == "s_waitcnt vmcnt(0)") == 1
        # descriptors: set up once at entry, re-pointed once (the next segment) in front of the last pair
        flags = [k for k, i in enumerate(ins) if i.endswith("0x00020000")]
        assert len(flags) == 4
        mfk = [k for k, i in enumerate(ins) if i.startswith(MF)]
        assert flags[1] < mfk[0] and mfk[(iters - 2) * 128 - 1] < flags[2] < flags[3] < mfk[(iters - 2) * 128] if iters > 2 else flags[3] < mfk[0]
    check_dma(pro, 32)
    cv = check_regs(seg + short + pro, wclob)
    assert min(cv) >= 48  # the compiler keeps v0 .. v47 at least
    # a half's reads never write the fragment set its own MFMAs consume
    half = []
    for i in seg:
        half.append(i)
        if i == "s_waitcnt lgkmcnt(0)":
            used = set()
            for j in half:
                m = re.match(MF + r" a\[\d+:\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\]", j)
                if m:
                    used |= {int(m.group(1)), int(m.group(2))}
            for j in half:
                m = re.match(RD + r" v\[(\d+):(\d+)\]", j)
                if m and used:
                    assert int(m.group(1)) // 4 * 4 not in used, j
            half = []
    # ---- paired form
    pclob = text[text.index("#define G4WP_CLOBBERS"):].split("\n", 1)[0]
    pseg, ppro = block("G4WP_ASM_SEG", "G4WP_CLOBBERS"), block("G4WP_ASM_PROLOGUE", "G4WP_ASM_SEG")
    pmf = [k for k, i in enumerate(pseg) if i.startswith(MF)]
    assert len(pmf) == 6 * 192 and all(pseg[k].endswith(", 0") for k in pmf[:64]) and sum(1 for k in pmf if pseg[k].endswith(", 0")) == 64
    assert sum(1 for i in pseg if i.startswith(RD)) == 48 + 6 * 64
    check_dma(pseg, 6 * 16)
    check_dma(ppro, 32)
    assert sum(1 for i in pseg if i == "s_waitcnt vmcnt(8)") == 6 and sum(1 for i in pseg if i == "s_barrier") == 7
    pcv = check_regs(pseg + ppro, pclob)
    assert min(pcv) >= 32
    for b in range(0, len(pmf), 64):
        used = set()
        for k in pmf[b:b + 64]:
            m = re.match(MF + r" a\[\d+:\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\]", pseg[k])
            used |= {int(m.group(1)), int(m.group(2))}
        for i in pseg[pmf[b]:pmf[b + 63] + 1]:
            m = re.match(RD + r" v\[(\d+):", i)
            if m:
                assert int(m.group(1)) // 4 * 4 not in used, (i, b // 64)


def test_conv4_generator_emits_a_consistent_instruction_stream(tmp_path):
    """The convolution form of csrc/gen_gemm4.py (conv4_kernel in conv8.hip: 256 x 192 tile, gathered A pieces).  Per iteration: 16 NI MFMAs on
    8 x NI accumulator blocks (NI = 6: 256 x 192 tile, 3: 256 x 96), 2 (8 + NI) fragment reads, 8 gathered A pieces -- each `buffer_load ... offen lds` preceded by its own M0 write and by the
    three instructions that build its offset from the row offset, the row's inverted tap mask and this K-tile's table entry -- and NI plain B
    pieces; one table read per iteration, behind the decode of the previous entry; the last pair takes the NEXT tile's rows; every named
    register lies inside the clobber list."""
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "g4.inc"
    subprocess.run([sys.executable, os.path.join(root, "instageo-e2e-geospatial-ml_amd", "csrc", "gen_gemm4.py"), str(out)], check=True)
    text = out.read_text()
    for NI in (6, 3):
        _check_conv4_blocks(text, NI)
    # ---- the paired split form (256 x 192 tile): three products of 48 MFMAs per 32-element K-tile on five quarter sets
    import re as _re
    pt = _re.findall(r'"([^"]*?)\\n\\t"', text[text.index("#define G4CP6_ASM_TILE"):text.index("#define G4CP6_CLOBBERS")])
    pclob = text[text.index("#define G4CP6_CLOBBERS"):].split("\n", 1)[0]
    MF = "v_mfma_f32_16x16x32_bf16"
    pmf = [k for k, i in enumerate(pt) if i.startswith(MF)]
    assert len(pmf) == 6 * 3 * 48 and all(pt[k].endswith(", 0") for k in pmf[:48]) and sum(1 for k in pmf if pt[k].endswith(", 0")) == 48
    assert sum(1 for i in pt if i.startswith("ds_read_b128")) == (8 + 8 + 6) + 6 * (6 + 8 + 6 + 8)
    assert sum(1 for i in pt if i.startswith("buffer_load_dwordx4")) == 6 * 8 and sum(1 for i in pt if i.startswith("global_load_lds_dwordx4")) == 6 * 6
    assert sum(1 for i in pt if i.startswith("ds_read_b32")) == 7 and sum(1 for i in pt if i == "s_barrier") == 7
    pcv = {int(x) for x in _re.findall(r'"v(\d+)"', pclob)}
    for i in pt:
        body = _re.sub(r"%\[[a-z0-9_]+\]", "", i)
        for lo, hi in _re.findall(r"\bv\[(\d+):(\d+)\]", body):
            assert set(range(int(lo), int(hi) + 1)) <= pcv, i
        for r in _re.findall(r"\bv(\d+)\b", body):
            assert int(r) in pcv, i
    for b in range(0, len(pmf), 48):  # a product's reads never write a quarter set its own MFMAs consume
        used = set()
        for k in pmf[b:b + 48]:
            m = _re.match(MF + r" a\[\d+:\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\]", pt[k])
            used |= {int(m.group(1)), int(m.group(2))}
        for i in pt[pmf[b]:pmf[b + 47] + 1]:
            m = _re.match(r"ds_read_b128 v\[(\d+):", i)
            if m:
                assert int(m.group(1)) not in used, (i, b // 48)


def _check_conv4_blocks(text, NI):
    import re

    tile = re.findall(r'"([^"]*?)\\n\\t"', text[text.index("#define G4C%d_ASM_TILE" % NI):text.index("#define G4C%d_CLOBBERS" % NI)])
    pro = re.findall(r'"([^"]*?)\\n\\t"', text[text.index("#define G4C%d_ASM_PROLOGUE" % NI):text.index("#define G4C%d_ASM_TILE" % NI)])
    clob = text[text.index("#define G4C%d_CLOBBERS" % NI):].split("\n", 1)[0]
    MF = "v_mfma_f32_16x16x32_bf16"
    mf = [i for i in tile if i.startswith(MF)]
    H = 8 * NI
    assert len(mf) == 6 * 2 * H and all(i.endswith(", 0") for i in mf[:H]) and sum(1 for i in mf if i.endswith(", 0")) == H
    for it in range(6):
        dst = sorted(int(re.match(MF + r" a\[(\d+):", i).group(1)) for i in mf[it * 2 * H:(it + 1) * 2 * H])
        assert dst == sorted([(mi * 8 + ni) * 4 for mi in range(8) for ni in range(NI)] * 2)
    assert sum(1 for i in tile if i.startswith("ds_read_b128")) == (8 + NI) + 6 * 2 * (8 + NI)
    assert sum(1 for i in tile if i.startswith("ds_read_b32")) == 1 + 6 and sum(1 for i in pro if i.startswith("ds_read_b32")) == 2
    ga = [k for k, i in enumerate(tile) if i.startswith("buffer_load_dwordx4")]
    gb = [k for k, i in enumerate(tile) if i.startswith("global_load_lds_dwordx4")]
    assert len(ga) == 6 * 8 and len(gb) == 6 * NI
    for k in ga + gb:
        assert tile[k - 2].startswith("s_add_u32 m0, ") and not tile[k - 1].startswith(("s_add_u32 m0", "buffer_load", "global_load")), tile[k - 2:k + 1]
    for n, k in enumerate(ga):  # the piece's offset register is written by bfe -> lshl_add -> add, in that order, before the issue
        reg = re.match(r"buffer_load_dwordx4 (v\d+),", tile[k]).group(1)
        ops3 = [i for i in tile[:k] if re.match(r"v_(bfe_u32|lshl_add_u32|add_u32) " + reg + ",", i)][-3:]
        assert [o.split()[0] for o in ops3] == ["v_bfe_u32", "v_lshl_add_u32", "v_add_u32"], (reg, ops3)
        nxt = n >= 4 * 8  # the last pair gathers the next tile's rows
        assert ("%[imn" in ops3[0]) == nxt and ("%[ron" in ops3[2]) == nxt, ops3
    assert sum(1 for i in tile if i == "s_waitcnt vmcnt(8)") == 6 and sum(1 for i in tile if i == "s_barrier") == 7
    cv = {int(x) for x in re.findall(r'"v(\d+)"', clob)}
    cs = {int(x) for x in re.findall(r'"s(\d+)"', clob)}
    assert {int(x) for x in re.findall(r'"a(\d+)"', clob)} == set(range(256)) and '"m0"' in clob and '"scc"' in clob
    for i in tile + pro:
        body = re.sub(r"%\[[a-z0-9_]+\]", "", i)
        for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", body):
            assert set(range(int(lo), int(hi) + 1)) <= cv, i
        for r in re.findall(r"\bv(\d+)\b", body):
            assert int(r) in cv, i
        for lo, hi in re.findall(r"\bs\[(\d+):(\d+)\]", body):
            assert set(range(int(lo), int(hi) + 1)) <= cs, i
        for r in re.findall(r"\bs(\d+)\b", body):
            assert int(r) in cs, i


def test_split_tensors_are_one_allocation_with_lo_above_hi():
    """ops.BT allocates hi and lo of a split tensor as ONE block, lo a fixed multiple of 256 bytes above hi, and slices taken at the same
    position keep that distance: the paired split-mode kernels (gemm8.hip / conv8.hip NSEG = 2) carry it in the lo lanes' 32-bit offsets."""
    import torch
    from instageo_amd.ops import BT

    for shape in [(7,), (5, 3), (1000, 768), (2, 3, 5, 48)]:
        for make in (BT.empty, BT.zeros):
            t = make(shape, True, "cpu")
            assert t.hi.shape == torch.Size(shape) and t.lo.shape == t.hi.shape and t.hi.is_contiguous() and t.lo.is_contiguous()
            d = t.lo.data_ptr() - t.hi.data_ptr()
            assert d >= t.hi.numel() * 2 and d % 256 == 0
            assert t.hi.untyped_storage().data_ptr() == t.lo.untyped_storage().data_ptr()
            flat = BT(t.hi.reshape(-1), t.lo.reshape(-1))
            assert flat.lo[3:].data_ptr() - flat.hi[3:].data_ptr() == d
        z = BT.zeros(shape, True, "cpu")
        assert not z.hi.any() and not z.lo.any()
        z.hi.fill_(1)
        assert not z.lo.any()  # the halves do not overlap
    assert BT.empty((4, 4), False, "cpu").lo is None


def test_operand_copy_reads_refuse_an_in_flight_all_gather():
    """Data parallel, deferred all-gather (distributed.ShardedGradSync, IG_DP_DEFER=1): the bf16 operand copy above the lowest pending bucket
    is being written by RCCL.  ``ParamStore.w`` -- the only accessor of operand views -- raises for a parameter at or above that offset
    (the ordering used to rest on every reader remembering to call ``wait_params``); below it, and with nothing pending, it does not."""
    from instageo_amd.model import PrithviSeg

    net = PrithviSeg(variant="prithvi_eo_tiny", temporal_step=1, num_classes=2, load_pretrained_weights=False, device="cpu")
    st = net.store
    names = list(st.entries)
    first, last = names[0], names[-1]
    boundary = st.entries[last].offset
    st.pending_from = lambda: boundary
    with pytest.raises(RuntimeError, match="still in flight"):
        st.w(last)
    st.shadow = type("S", (), {"hi": st.flat.to(torch.bfloat16), "lo": None})()  # a stand-in operand copy (the real one is made by a HIP kernel)
    assert st.w(first).hi.numel() == st.entries[first].numel  # below the pending bucket: readable
    st.pending_from = lambda: None
    assert st.w(last).hi.numel() == st.entries[last].numel


def test_gemm4_generator_emits_a_consistent_instruction_stream(tmp_path):
    """csrc/gen_gemm4.py writes the hand-scheduled K-loop of gemm4.hip (one inline-asm block per tile, registers assigned by hand).  Structural
    invariants of what it emits, checked on the generated text (no GPU, no compiler):
    * the tile block holds 6 iterations (first / middle / last pair) of 128 MFMAs, every accumulator block a[4i : 4i+3] is the destination of
      exactly two MFMAs per iteration, and only the 64 MFMAs of the very first half start from the inline constant 0;
    * per iteration 32 fragment reads (+ 16 at the tile's entry), 16 LDS-DMA issues each preceded by its own M0 write with one other instruction
      between them, one `vmcnt(8)` + one `s_barrier` (+ the entry's `vmcnt(0)` + barrier);
    * every explicitly named VGPR / SGPR / AGPR lies inside the clobber list, M0 and SCC are declared, and a fragment register is never the
      destination of a read in the half whose MFMAs consume it."""
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "g4.inc"
    subprocess.run([sys.executable, os.path.join(root, "instageo-e2e-geospatial-ml_amd", "csrc", "gen_gemm4.py"), str(out)], check=True)
    text = out.read_text()
    tile = text[text.index("#define G4_ASM_TILE"):text.index("#define G4_CLOBBERS")]
    pro = text[text.index("#define G4_ASM_PROLOGUE"):text.index("#define G4_ASM_TILE")]
    clob = text[text.index("#define G4_CLOBBERS"):].split("\n", 1)[0]
    ins = re.findall(r'"([^"]*?)\\n\\t"', tile)
    mf = [i for i in ins if i.startswith("v_mfma_f32_16x16x32_bf16")]
    assert len(mf) == 6 * 128
    assert sum(1 for i in mf if i.endswith(", 0")) == 64 and all(i.endswith(", 0") for i in mf[:64])
    for it in range(6):
        dst = [re.match(r"v_mfma_f32_16x16x32_bf16 a\[(\d+):(\d+)\]", i).group(1) for i in mf[it * 128:(it + 1) * 128]]
        assert sorted(map(int, dst)) == sorted([4 * b for b in range(64)] * 2)
    assert sum(1 for i in ins if i.startswith("ds_read_b128")) == 16 + 6 * 32
    dma = [k for k, i in enumerate(ins) if i.startswith("global_load_lds_dwordx4")]
    assert len(dma) == 6 * 16
    for k in dma:  # M0 write two instructions earlier, something else in between (the wait state M0 needs)
        assert ins[k - 2].startswith("s_add_u32 m0, ") and not ins[k - 1].startswith(("s_add_u32 m0", "global_load_lds")), ins[k - 2:k + 1]
    assert sum(1 for i in ins if i == "s_waitcnt vmcnt(8)") == 6 and sum(1 for i in ins if i == "s_waitcnt vmcnt(0)") == 1
    assert sum(1 for i in ins if i == "s_barrier") == 7
    pins = re.findall(r'"([^"]*?)\\n\\t"', pro)
    assert sum(1 for i in pins if i.startswith("global_load_lds_dwordx4")) == 32  # K-tiles 0 and 1 of the workgroup's first tile
    cv = {int(x) for x in re.findall(r'"v(\d+)"', clob)}
    cs = {int(x) for x in re.findall(r'"s(\d+)"', clob)}
    ca = {int(x) for x in re.findall(r'"a(\d+)"', clob)}
    assert ca == set(range(256)) and '"m0"' in clob and '"scc"' in clob and '"memory"' in clob
    for i in ins + pins:
        body = re.sub(r"%\[[a-z0-9_]+\]", "", i)
        for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", body):
            assert set(range(int(lo), int(hi) + 1)) <= cv, i
        for r in re.findall(r"\bv(\d+)\b", body):
            assert int(r) in cv, i
        for lo, hi in re.findall(r"\bs\[(\d+):(\d+)\]", body):
            assert set(range(int(lo), int(hi) + 1)) <= cs, i
        for r in re.findall(r"\bs(\d+)\b", body):
            assert int(r) in cs, i
    # ---- the paired form (split precision mode): three products per K-tile, five quarter sets
    ptile = text[text.index("#define G4P_ASM_TILE"):text.index("#define G4P_CLOBBERS")]
    pclob = text[text.index("#define G4P_CLOBBERS"):].split("\n", 1)[0]
    pin = re.findall(r'"([^"]*?)\\n\\t"', ptile)
    pmf = [k for k, i in enumerate(pin) if i.startswith("v_mfma_f32_16x16x32_bf16")]
    assert len(pmf) == 6 * 192 and all(pin[k].endswith(", 0") for k in pmf[:64]) and sum(1 for k in pmf if pin[k].endswith(", 0")) == 64
    assert sum(1 for i in pin if i.startswith("ds_read_b128")) == 24 + 6 * 32 and sum(1 for i in pin if i.startswith("global_load_lds_dwordx4")) == 6 * 16
    assert sum(1 for i in pin if i == "s_waitcnt vmcnt(8)") == 6 and sum(1 for i in pin if i == "s_barrier") == 7
    pcv = {int(x) for x in re.findall(r'"v(\d+)"', pclob)}
    for i in pin:
        body = re.sub(r"%\[[a-z0-9_]+\]", "", i)
        for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", body):
            assert set(range(int(lo), int(hi) + 1)) <= pcv, i
        for r in re.findall(r"\bv(\d+)\b", body):
            assert int(r) in pcv, i
    for b in range(0, len(pmf), 64):  # a product's reads never write a quarter set its own MFMAs consume
        lo_k, hi_k = pmf[b], pmf[b + 63]
        used = set()
        for k in pmf[b:b + 64]:
            m = re.match(r"v_mfma_f32_16x16x32_bf16 a\[\d+:\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\]", pin[k])
            used |= {int(m.group(1)), int(m.group(2))}
        for i in pin[lo_k:hi_k + 1]:
            m = re.match(r"ds_read_b128 v\[(\d+):\d+\]", i)
            if m:
                assert int(m.group(1)) not in used, (i, b // 64)
    # a half's reads never write the fragment set its MFMAs consume: split the stream at the waits that end a half
    half, halves = [], []
    for i in ins:
        half.append(i)
        if i == "s_waitcnt lgkmcnt(0)":
            halves.append(half)
            half = []
    for h in halves:
        used = set()
        for i in h:
            m = re.match(r"v_mfma_f32_16x16x32_bf16 a\[\d+:\d+\], v\[(\d+):\d+\], v\[(\d+):\d+\]", i)
            if m:
                used |= {int(m.group(1)), int(m.group(2))}
        for i in h:
            m = re.match(r"ds_read_b128 v\[(\d+):\d+\]", i)
            if m and used:
                assert int(m.group(1)) not in used, (i, sorted(used)[:4])
