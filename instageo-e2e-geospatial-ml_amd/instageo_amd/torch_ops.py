"""PyTorch custom-op registration of the HIP library: namespace ``instageo_mi355x`` (north_star "behind
PyTorch-ROCm custom ops").

Two layers, both on top of the ctypes C-ABI binding (:mod:`instageo_amd._lib`; ``include/instageo_hip.h`` stays the single
source of truth -- the schemas below are GENERATED from its prototypes):

* **raw ops**, one per C-ABI entry point: ``torch.ops.instageo_mi355x.<entry without ig_>``.  Pointer parameters become
  ``Tensor?`` arguments (non-``const`` pointers are declared mutated, ``Tensor(a!)?``), scalars keep their C meaning, the
  trailing ``stream`` is torch's current HIP stream.  They return nothing (every output is a caller-allocated, mutated
  argument -- the reference's ATen ops with ``out=``), so the fake / meta implementation is a no-op and the ops are visible to
  the dispatcher, ``torch.compile`` graphs (as opaque mutating calls) and ``torch.library.opcheck``.
* **functional, differentiable ops** for the Block linears and LayerNorm with ``torch.library.register_autograd``:
  ``linear(x, w, bias, act)``, ``layer_norm(x, gamma, beta, eps)`` -- bf16 activations/weights, fp32 residual-stream input for
  LayerNorm -- whose backward formulas call the dgrad / wgrad / layernorm_bwd raw ops.
* **the whole network as one op**: ``prithvi_seg(img, params[], handle, training, save, features) -> (logits, features)`` with
  ``prithvi_seg_backward`` behind ``register_autograd`` -- what ``PrithviSeg.forward`` dispatches through (model.py).  The engine
  (workspaces, saved activations, BatchNorm running statistics) is found through the integer ``handle``; the parameters travel as a
  tensor list so that autograd sees the dependence.

There is no CPU implementation: the ops are registered for the CUDA (HIP) dispatch key only.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib

NAMESPACE = "instageo_mi355x"
_SCALAR_SCHEMA = {"int": "int", "long": "int", "unsigned": "int", "float": "float", "double": "float"}
_SKIP = {"ig_last_error", "ig_last_kernel", "ig_note_reset", "ig_last_grid", "ig_version", "ig_header_stamp", "ig_device_info",
         "ig_set_reserved_cus", "ig_get_reserved_cus", "ig_set_deterministic", "ig_get_deterministic", "ig_det_fold", "ig_det_fold_ranges",
         "ig_linear_wgrad_group", "ig_conv3x3_fwd_stats", "ig_conv3x3_cls_fwd"}  # host-side queries: no tensors; the grouped launch takes HOST arrays of device
# pointers; ig_conv3x3_fwd_stats / ig_conv3x3_cls_fwd report through a HOST int (ig_conv3x3_fwd + ig_bn_relu_fwd / ig_classifier_fwd are the
# op-level equivalents)


def parse_prototypes(path: str = _lib.HEADER_PATH) -> Dict[str, List[Tuple[str, str]]]:
    """{entry point: [(C type, parameter name), ...]} for every ``int ig_*(...)`` prototype of the header."""
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    out: Dict[str, List[Tuple[str, str]]] = {}
    for m in re.finditer(r"\bint\s+(ig_\w+)\s*\(([^)]*)\)\s*;", text):
        name, args = m.group(1), m.group(2).strip()
        params: List[Tuple[str, str]] = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                params.append((mm.group(1).strip(), mm.group(2)))
        out[name] = params
    return out


def schema_of(name: str, params: List[Tuple[str, str]]) -> Tuple[str, List[Tuple[str, str, bool]]]:
    """-> (schema string without the op name, [(kind, parameter, mutated)]) ; kind in {"tensor", "int", "float"}."""
    parts, kinds = [], []
    alias = iter("abcdefghijklmnopqrstuvwxyz")
    for ctype, pname in params:
        if pname == "stream":
            continue
        if "*" in ctype:
            mutated = not ctype.startswith("const")
            parts.append(f"Tensor({next(alias)}!)? {pname}" if mutated else f"Tensor? {pname}")
            kinds.append(("tensor", pname, mutated))
        else:
            t = _SCALAR_SCHEMA[ctype]
            parts.append(f"{t} {pname}")
            kinds.append((t, pname, False))
    return "(" + ", ".join(parts) + ") -> ()", kinds


_LIBRARY: Optional[torch.library.Library] = None
RAW_OPS: Dict[str, str] = {}  # op name -> schema


def _make_impl(entry: str, kinds):
    def impl(*args):
        call = []
        for (kind, _, _), a in zip(kinds, args):
            if kind == "tensor":
                if a is None:
                    call.append(None)
                else:
                    if not a.is_cuda or not a.is_contiguous():
                        raise _lib.HipLibraryError(f"{NAMESPACE}::{entry[3:]}: tensors must be contiguous HIP tensors")
                    call.append(a.data_ptr())
            else:
                call.append(a)
        call.append(torch.cuda.current_stream().cuda_stream)
        _lib.call(entry, *call)

    return impl


def register() -> Dict[str, str]:
    """Define and implement every op once per process (idempotent); returns {op name: schema}."""
    global _LIBRARY
    if _LIBRARY is not None:
        return RAW_OPS
    lib = torch.library.Library(NAMESPACE, "DEF")
    for entry, params in parse_prototypes().items():
        if entry in _SKIP or not any("*" in t for t, n in params if n != "stream"):
            continue
        schema, kinds = schema_of(entry, params)
        op = entry[3:]
        lib.define(op + schema)
        lib.impl(op, _make_impl(entry, kinds), "CUDA")
        torch.library.register_fake(f"{NAMESPACE}::{op}", lambda *a, **k: None, lib=lib)
        RAW_OPS[op] = schema
    _register_functional(lib)
    _register_network(lib)
    _LIBRARY = lib
    return RAW_OPS


# ---------------------------------------------------------------------------------------------------------------------
# functional, differentiable ops
# ---------------------------------------------------------------------------------------------------------------------
def _register_functional(lib: torch.library.Library) -> None:
    ns = torch.ops.instageo_mi355x

    # y = act(x @ w^T + bias): x (M, K) bf16, w (N, K) bf16, bias (N) f32 | None, act 0 none / 1 exact GELU.
    # Returns (y, dact): dact = gelu'(pre-activation) (empty when act == 0), the factor backward applies.
    lib.define("linear(Tensor x, Tensor w, Tensor? bias, int act) -> (Tensor, Tensor)")

    def linear_impl(x, w, bias, act):
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
        dact = torch.empty((M, N) if act == 1 else (0,), dtype=torch.bfloat16, device=x.device)
        ns.linear_fwd(x, None, w, None, bias, y, None, dact if act == 1 else None, None, M, N, K, act)
        return y, dact

    lib.impl("linear", linear_impl, "CUDA")

    def linear_fake(x, w, bias, act):
        return x.new_empty((x.shape[0], w.shape[0])), x.new_empty((x.shape[0], w.shape[0]) if act == 1 else (0,))

    torch.library.register_fake(f"{NAMESPACE}::linear", linear_fake, lib=lib)

    # (dx, dw, dbias) of linear for an upstream dy: dy * dact first when act == 1
    lib.define("linear_backward(Tensor dy, Tensor x, Tensor w, Tensor dact, int act, bool need_bias) -> (Tensor, Tensor, Tensor)")

    def linear_backward_impl(dy, x, w, dact, act, need_bias):
        M, K = x.shape
        N = w.shape[0]
        if act == 1:
            dy = (dy.float() * dact.float()).to(torch.bfloat16)
        dy = dy.contiguous()
        dx = torch.empty((M, K), dtype=torch.bfloat16, device=x.device)
        ns.linear_dgrad(dy, None, w, None, dx, None, None, None, None, M, N, K, 0)
        dw = torch.zeros((N, K), dtype=torch.float32, device=x.device)
        ns.linear_wgrad(dy, None, x, None, dw, M, N, K)
        db = torch.zeros((N if need_bias else 0,), dtype=torch.float32, device=x.device)
        if need_bias:
            ns.colsum(dy, None, db, M, N)
        return dx, dw, db

    lib.impl("linear_backward", linear_backward_impl, "CUDA")
    torch.library.register_fake(
        f"{NAMESPACE}::linear_backward",
        lambda dy, x, w, dact, act, need_bias: (x.new_empty(x.shape), w.new_empty(w.shape, dtype=torch.float32),
                                                w.new_empty((w.shape[0] if need_bias else 0,), dtype=torch.float32)),
        lib=lib)  # fmt: skip

    def linear_setup(ctx, inputs, output):
        x, w, bias, act = inputs
        ctx.save_for_backward(x, w, output[1])
        ctx.act, ctx.has_bias = act, bias is not None

    def linear_bwd(ctx, dy, _ddact):
        x, w, dact = ctx.saved_tensors
        dx, dw, db = ns.linear_backward(dy.contiguous(), x, w, dact, ctx.act, ctx.has_bias)
        return dx, dw.to(w.dtype), (db if ctx.has_bias else None), None

    torch.library.register_autograd(f"{NAMESPACE}::linear", linear_bwd, setup_context=linear_setup, lib=lib)

    # nn.LayerNorm over the last dim: x (M, D) f32 residual stream -> (y bf16, mean f32, rstd f32)
    lib.define("layer_norm(Tensor x, Tensor gamma, Tensor beta, float eps) -> (Tensor, Tensor, Tensor)")

    def ln_impl(x, gamma, beta, eps):
        M, D = x.shape
        y = torch.empty((M, D), dtype=torch.bfloat16, device=x.device)
        mean = torch.empty((M,), dtype=torch.float32, device=x.device)
        rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
        ns.layernorm_fwd(x, gamma, beta, y, None, mean, rstd, M, D, eps, 0, 0, 0)
        return y, mean, rstd

    lib.impl("layer_norm", ln_impl, "CUDA")
    torch.library.register_fake(
        f"{NAMESPACE}::layer_norm",
        lambda x, gamma, beta, eps: (x.new_empty(x.shape, dtype=torch.bfloat16), x.new_empty((x.shape[0],)), x.new_empty((x.shape[0],))),
        lib=lib)  # fmt: skip
    lib.define("layer_norm_backward(Tensor dy, Tensor x, Tensor mean, Tensor rstd, Tensor gamma) -> (Tensor, Tensor, Tensor)")

    def ln_bwd_impl(dy, x, mean, rstd, gamma):
        M, D = x.shape
        dx = torch.empty_like(x)
        dg = torch.zeros((D,), dtype=torch.float32, device=x.device)
        db = torch.zeros((D,), dtype=torch.float32, device=x.device)
        ns.layernorm_bwd(dy.contiguous(), None, x, mean, rstd, gamma, dx, 0, None, None, dg, db, None, M, D, 0, 0, 0)
        return dx, dg, db

    lib.impl("layer_norm_backward", ln_bwd_impl, "CUDA")
    torch.library.register_fake(
        f"{NAMESPACE}::layer_norm_backward",
        lambda dy, x, mean, rstd, gamma: (x.new_empty(x.shape), gamma.new_empty(gamma.shape), gamma.new_empty(gamma.shape)), lib=lib)

    def ln_setup(ctx, inputs, output):
        x, gamma, _, _ = inputs
        ctx.save_for_backward(x, gamma, output[1], output[2])

    def ln_bwd(ctx, dy, _dm, _dr):
        x, gamma, mean, rstd = ctx.saved_tensors
        dx, dg, db = ns.layer_norm_backward(dy.to(torch.bfloat16), x, mean, rstd, gamma)
        return dx, dg, db, None

    torch.library.register_autograd(f"{NAMESPACE}::layer_norm", ln_bwd, setup_context=ln_setup, lib=lib)


# ---------------------------------------------------------------------------------------------------------------------
# the whole network as one op (PrithviSeg.forward dispatches through it)
# ---------------------------------------------------------------------------------------------------------------------
def _register_network(lib: torch.library.Library) -> None:
    from . import model as M

    ns = torch.ops.instageo_mi355x
    # img (B, C, T, H, W) f32 [or (B, C, H, W) when T == 1]; params: the module's parameters in flat (= forward) order -- read through
    # the engine's own flat buffer, listed here for autograd; handle: model.network_of; training: BatchNorm batch statistics + dropout
    # (nn.Module.train()); save: keep the activations for prithvi_seg_backward; features: also return reshaped_features (B, D*T, 14, 14)
    # (an empty tensor otherwise).  Hidden state behind the handle: workspaces, the saved activations of the latest save=True call,
    # the BatchNorm running statistics (updated when training), the dropout counter.
    # Third output: the engine's forward GENERATION as a 1-element int64 CPU tensor.  It is a real op output -- not Python state read in
    # setup_context -- so that a traced graph (torch.compile / AOTAutograd, where setup_context runs once on fake tensors) hands the
    # backward the generation of the forward that actually ran, not a constant baked in at trace time.
    lib.define("prithvi_seg(Tensor img, Tensor[] params, int handle, bool training, bool save, bool features) -> (Tensor, Tensor, Tensor)")

    def seg_impl(img, params, handle, training, save, features):
        net = M.network_of(handle)
        eng = net.engine
        eng.mark_params_changed()  # foreign optimizers write into the fp32 views directly: always refresh the bf16 operands
        logits = eng.forward(img, training, save=save or features, update_running=True)
        feats = eng.features_nchw() if features else logits.new_empty((0,))
        return logits, feats, torch.tensor([eng._generation], dtype=torch.int64)

    lib.impl("prithvi_seg", seg_impl, "CUDA")

    def seg_fake(img, params, handle, training, save, features):
        net = M.network_of(handle)
        cfg = net.engine.geometry(int(img.shape[-1]))
        B = img.shape[0]
        logits = img.new_empty((B, cfg.num_classes, cfg.out_size, cfg.out_size), dtype=torch.float32)
        shape = (B, cfg.embed_dim * cfg.num_frames, cfg.grid, cfg.grid) if features else (0,)
        return logits, img.new_empty(shape, dtype=torch.float32), torch.empty((1,), dtype=torch.int64, device="cpu")

    torch.library.register_fake(f"{NAMESPACE}::prithvi_seg", seg_fake, lib=lib)

    # gradients of the parameters for an upstream dlogits, from the activations the forward call of ``generation`` saved; an empty
    # tensor for a parameter that does not require a gradient
    lib.define("prithvi_seg_backward(Tensor dlogits, Tensor[] params, int handle, Tensor generation) -> Tensor[]")

    def seg_bwd_impl(dlogits, params, handle, generation):
        net = M.network_of(handle)
        eng, store = net.engine, net.store
        g = store.ensure_grad()
        g.zero_()
        eng.backward(dlogits.contiguous().float(), generation=int(generation.item()))  # a CPU tensor: no device synchronisation
        return [store.entries[name].api_view(g).clone() if p.requires_grad else p.new_empty((0,)) for name, p in net._flat_params()]

    lib.impl("prithvi_seg_backward", seg_bwd_impl, "CUDA")
    torch.library.register_fake(
        f"{NAMESPACE}::prithvi_seg_backward",
        lambda dlogits, params, handle, generation: [p.new_empty(p.shape if p.requires_grad else (0,)) for p in params], lib=lib)

    def seg_setup(ctx, inputs, output):
        img, params, handle, training, save, features = inputs
        ctx.handle = handle
        ctx.req = [p.requires_grad for p in params]
        ctx.set_materialize_grads(False)  # an unused output arrives as None, so a gradient INTO the features can be told from no gradient
        ctx.save_for_backward(*params, output[2])

    def seg_bwd(ctx, dlogits, dfeats, _dgen):
        # The engine propagates a gradient from the logits only.  A loss built on the features output (return_features=True) used to
        # train nothing through it, silently: refuse instead.
        if dfeats is not None:
            raise RuntimeError("instageo_mi355x::prithvi_seg: the features output is not differentiable (the engine back-propagates from "
                               "the logits only); detach() it, or build the loss on the logits")
        if dlogits is None:
            raise RuntimeError("instageo_mi355x::prithvi_seg: backward without a gradient for the logits")
        *params, gen = ctx.saved_tensors
        grads = ns.prithvi_seg_backward(dlogits, list(params), ctx.handle, gen)
        return None, [g if r else None for g, r in zip(grads, ctx.req)], None, None, None, None

    torch.library.register_autograd(f"{NAMESPACE}::prithvi_seg", seg_bwd, setup_context=seg_setup, lib=lib)
