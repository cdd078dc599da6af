"""Data-parallel integration (-m gpu): two ranks (gloo backend, both on cuda:0 because the GPU box has one device) run
the real fused training step with bucketed gradient averaging; the result must equal one process that computes both
ranks' gradients separately, averages them and applies the same AdamW step (Lightning-DDP semantics: mean of the
per-rank gradients, rank-local BatchNorm)."""
import os
import socket

import pytest
import numpy as np
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_module(device="cuda:0"):
    from instageo_amd.segmentation import PrithviSegmentationModule
    from oracle import prithvi_oracle as O
    from oracle.cases import case_config

    cfg = case_config("tiny_t1_c2")
    mod = PrithviSegmentationModule(freeze_backbone=False, load_pretrained_weights=False, num_classes=2, model_name="prithvi_eo_tiny",
                                    class_weights=[1, 3], ignore_index=-1, learning_rate=1e-3, precision="bf16x3", device=device)
    mod.net.load_state_dict(O.make_state_dict(cfg, seed=1042))
    mod.net.cfg.drop_p = 0.0
    return cfg, mod


def _batch(cfg, rank):
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(2, 6, 1, 224, 224, generator=g).cuda()
    y = torch.randint(0, 2, (2, 224, 224), generator=g)
    y[0, :10] = -1
    return x, y.cuda()


def _worker(rank, world, port, q, backend="gloo", mode="zero1", force=False):
    import sys

    os.environ["IG_DP_MODE"] = mode
    if force:
        os.environ["IG_DIST_FORCE"] = "1"

    sys.path[:0] = [ROOT, os.path.join(ROOT, "instageo-e2e-geospatial-ml_amd")]
    local = str(rank) if backend == "nccl" else "0"  # RCCL: one rank per GPU; gloo: both ranks share cuda:0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=local,
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from instageo_amd import distributed as D

    try:
        _, local_rank, _ = D.init_from_env(backend=backend)
        torch.cuda.set_device(local_rank)
        cfg, mod = _make_module(f"cuda:{local_rank}")
        if rank == 1:  # replicas start different: attach_data_parallel must broadcast rank 0's weights
            mod.net.store.flat.mul_(1.01)
        sync = D.attach_data_parallel(mod, bucket_bytes=1 << 20)
        assert isinstance(sync, D.ShardedGradSync if mode == "zero1" else D.GradSync)
        assert (mod.optimizer().m is None) == (mode == "zero1")  # sharded: no replicated moment buffers
        x, y = _batch(cfg, rank)
        for _ in range(2):
            mod.fused_train_step(x, y)
        torch.cuda.synchronize()
        if mode == "zero1":
            # the all-gathers of the operand copy are left in flight for the next forward pass to wait for, Block by Block
            # (engine.param_wait); a reader of the copy outside the engine waits for them itself
            deferred = len(sync._pending)
            assert world == 1 or os.environ.get("IG_DP_DEFER") == "0" or deferred > 0
            sync.wait_params()
        sh = mod.net.store.shadow
        shadow_sum = float(sh.hi.float().double().sum().item() + (0.0 if sh.lo is None else sh.lo.float().double().sum().item()))
        if mode == "zero1":
            # between checkpoints only the bf16 operand copy is exchanged: the fp32 masters of the slices the other rank owns are
            # stale until sync_master_params() (collective) completes them
            stale = mod.net.store.flat.detach().clone()
            assert not sync.master_complete
            mod.sync_master_params()
            assert sync.master_complete and (world == 1 or not torch.equal(stale, mod.net.store.flat))  # (one rank owns every slice)
        flat = mod.net.store.flat.detach().cpu()
        # numpy (pickled by value): a torch tensor would travel as a shared-memory fd that dies with this process
        from instageo_amd import ops

        q.put((rank, flat.numpy().copy(), len(sync.launched), int(mod.train_metrics.matrix.sum()), ops.reserved_cus(), shadow_sum))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc(), 0, 0, 0, 0.0))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("backend,mode", [("gloo", "zero1"), ("gloo", "allreduce"), ("nccl", "zero1")])
def test_two_rank_fused_training_equals_manual_gradient_mean(backend, mode):
    """backend "nccl" (= RCCL, one rank per GPU) runs where the box has >= 2 devices and is skipped on the 1-GPU test box; the
    gloo variants exercise the same bucketing / hook / AdamW-scale code with both ranks on cuda:0.  mode "zero1" (the default):
    reduce-scatter + AdamW on the owned half of every bucket + all-gather; "allreduce": the replicated optimizer."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL variant needs two GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert not isinstance(r[1], str), r[1]
    flat0, flat1 = torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])
    assert torch.equal(flat0, flat1), "replicas diverged"
    assert res[0][5] == res[1][5], "the bf16 operand copies of the replicas differ"
    assert res[0][2] >= 2, "expected several gradient buckets per step"
    assert res[0][4] == 8 and res[1][4] == 8, "attach_data_parallel leaves 8 CUs to the collective kernels (distributed.DEFAULT_RESERVED_CUS)"
    # single-process restatement: per-rank grads (rank-local BN), mean, one AdamW step -- twice
    from instageo_amd import ops

    cfg, mod = _make_module()
    twin_cfg, twin = _make_module()  # rank-1 replica: BN running stats evolve per rank, weights are shared
    opt = mod.optimizer()
    for _ in range(2):
        grads = []
        for rank, m in ((0, mod), (1, twin)):
            m.net.store.flat.copy_(mod.net.store.flat)
            m.net.params_changed()
            x, y = _batch(cfg, rank)
            eng = m.net.engine
            logits = eng.forward(x, training=True, save=True)
            stats = torch.zeros(2, dtype=torch.float64, device="cuda")
            dlog = torch.empty_like(logits)
            ops.ce_loss(logits, y, m._weights(), -1, stats, dlog)
            g = m.net.store.ensure_grad()
            g.zero_()
            eng.backward(dlog, count=stats)
            grads.append(g.clone())
        mod.net.store.grad.copy_((grads[0] + grads[1]) / 2)
        opt.step(grads_in_flat=True)
    ref = mod.net.store.flat.detach().cpu()
    # fp32 wgrad atomics make gradients order-dependent in the last bits and Adam's first steps move every weight by
    # ~lr * sign(g): a noise-level gradient may flip sign, so single weights may differ by up to 2 steps * 2 * lr.
    from oracle import prithvi_oracle as O

    init = torch.zeros_like(ref)
    _, fresh = _make_module()
    init = fresh.net.store.flat.detach().cpu()
    d_dp, d_ref = flat0 - init, ref - init
    cos = torch.nn.functional.cosine_similarity(d_dp.reshape(1, -1), d_ref.reshape(1, -1)).item()
    diff = (flat0 - ref).abs()
    print(f"update cosine {cos:.6f}; max diff {diff.max().item():.2e}; mean diff {diff.mean().item():.2e}")
    assert cos >= 0.999 and diff.max().item() <= 4.5e-3 and diff.mean().item() <= 2e-5


def test_deferred_all_gather_equals_the_blocking_form_bit_for_bit(monkeypatch):
    """``IG_DP_DEFER=1`` (default: the all-gathers of the bf16 operand copy are left in flight and waited for by the next forward pass, Block
    by Block) against ``IG_DP_DEFER=0`` (waited for inside ``optimizer.step()``): two ranks (gloo, sharing cuda:0 -- the transport this pool
    offers; on RCCL ``bench.py --gpus N`` makes the same comparison in its pre-flight) run the real fused training step twice under each
    setting.  The completed fp32 parameters and the operand copy must be identical between the two forms, on both ranks."""
    ctx = mp.get_context("spawn")
    out = {}
    for defer in ("1", "0"):
        monkeypatch.setenv("IG_DP_DEFER", defer)  # inherited by the spawned workers
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, q, "gloo", "zero1")) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
        for r in res:
            assert not isinstance(r[1], str), r[1]
        out[defer] = res
    for rank in (0, 1):
        a, b = out["1"][rank], out["0"][rank]
        assert np.array_equal(a[1], b[1]), f"rank {rank}: fp32 parameters differ between the deferred and the blocking all-gather"
        assert a[5] == b[5], f"rank {rank}: bf16 operand copies differ between the deferred and the blocking all-gather"


@pytest.mark.parametrize("mode", ["zero1", "allreduce"])
def test_single_rank_rccl_preflight(mode):
    """The collectives of the data-parallel step on RCCL itself (backend "nccl"), with the ONE device this box has: ``IG_DIST_FORCE=1``
    initialises a one-rank process group and keeps every collective of the path in place (parameter broadcast, bucketed reduce-scatter /
    all-reduce launched from backward, sharded AdamW, all-gather of the bf16 operand copy, the compact all-reduce of the small fp32
    parameters, ``sync_master_params``) -- each degenerates to a copy, so the result must equal the plain single-process step.  What this
    screens is what two gloo ranks cannot: dtypes, alignment and stream semantics RCCL accepts for exactly our call pattern."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), q, "nccl", mode, True))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=60)
    assert not isinstance(res[1], str), res[1]
    assert res[2] >= 2, "expected several gradient buckets per step"
    cfg, mod = _make_module()
    x, y = _batch(cfg, 0)
    for _ in range(2):
        mod.fused_train_step(x, y)
    torch.cuda.synchronize()
    ref = mod.net.store.flat.detach().cpu()
    got = torch.from_numpy(res[1])
    _, fresh = _make_module()
    init = fresh.net.store.flat.detach().cpu()
    cos = torch.nn.functional.cosine_similarity((got - init).reshape(1, -1), (ref - init).reshape(1, -1)).item()
    diff = (got - ref).abs()
    print(f"update cosine {cos:.6f}; max diff {diff.max().item():.2e}; mean diff {diff.mean().item():.2e}")
    assert cos >= 0.999 and diff.max().item() <= 4.5e-3 and diff.mean().item() <= 2e-5
