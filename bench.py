#!/usr/bin/env python3
"""Headline benchmark: Prithvi-100M segmentation training step (fwd + CE + bwd + AdamW) in chips/s.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (for N>1 launched by torch.distributed.run, one rank per GPU over
RCCL).  Prints ONE JSON line on rank 0.  Workload = BASELINE.json configs[1]: Prithvi-100M fine-tune on Sen1Floods11-shaped
synthetic chips (6 bands, T=1, 224x224, 2 classes, class weights [1,3], ignore_index -1, dropout on), bf16 MFMA, per-GPU batch
``--batch`` (weak scaling).  A step = K0 normalise of a resident int16 batch -> forward -> loss/metrics -> backward ->
(gradient all-reduce) -> AdamW.

The line is kept COMPACT (< 2000 characters: the driver's record keeps only a tail of stdout).  On it, besides the contract keys:
  config             workload + the other legs' headline numbers: encoder_fwd_mfma_frac / encoder_fwd_ms (the encoder forward
                     alone: patch embed + L blocks + final LayerNorm -- the north-star target is stated on it),
                     inference_chips_per_s (K0 + forward + argmax int8), parity_mode_chips_per_s / _inference (the SAME workload
                     in bf16x3, the mode that meets the 1e-3 logits bar), yaml_b16 / yaml_t3_b8 / t3_b72 _chips_per_s (N = 1: short
                     legs at the reference YAMLs' batches and at configs[2]'s shape), tile_windows_per_s (BASELINE.json configs[3]: sliding
                     window over a resident 10980^2 int16 tile, window gather included; N > 1: windows sharded by rank, final
                     RCCL gather of the int8 maps included, plus the per-GPU rate without the gather), whole_step_mfma_frac
  roofline           the dominant KERNEL of the timed region (per-step time = launches/step x average launch): algorithmic FLOPs
                     of its launches / HIP-event time, against the dense bf16 MFMA peak; `traffic` from the committed PMC passes
  cpu_baseline       N=1: the CPU oracle (kind "port") on the host cores: the train step (16 threads: the best count of the
                     sweep on the 256-core boxes, stated in `cores`) and the configs[0] forward (B=4)
  dist               N>1: ranks, backend, reserved CUs, all-reduce milliseconds per step (instrumented extra step)
  detail             path of the JSON file with everything else (written by rank 0, default profiles/bench_detail_*.json):
                     roofline_timed_region / roofline_kernels (every MFMA kernel keyed by the name rocprofv3 prints), hbm_ops (HBM-bound
                     entry points against 8 TB/s), host_input (PCIe-inclusive rates), per-bucket all-reduce times, full notes

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (a ``torch.distributed.run`` child process,
before anything touches the GPU) and exits with the child's code.  ``--dry-run`` exercises launch, rendezvous, barriers,
max-over-ranks timing and the JSON line without the HIP library (CPU boxes: IG_DIST_BACKEND=gloo).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "instageo-e2e-geospatial-ml_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MEAN = [0.14245495, 0.13921481, 0.12434631, 0.31420089, 0.20743526, 0.12046503]  # sen1floods11.yaml:33-34
STD = [0.04036231, 0.04186983, 0.05267646, 0.0822221, 0.06834774, 0.05294205]
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0     # HBM3E (MI355X_MICROARCH.md)


def flop_per_chip_fwd(D: int, L: int, T: int, ncls: int):
    """blocks 24*N*D^2 + 4*N^2*D, patch embed, 4 head stages, classifier -> (total, encoder)."""
    N = 1 + 196 * T
    enc = L * (24.0 * N * D * D + 4.0 * N * N * D) + 2.0 * (196 * T) * 1536 * D
    head = 0.0
    for i in range(4):
        cin, hin = D * T / 2**i, 14 * 2**i
        cout = cin / 2
        head += 2 * cin * cout * 9 * hin**2 + 2 * cout * cout * 9 * (2 * hin) ** 2
    return enc + head + 2.0 * (D * T / 16) * ncls * 224**2, enc


GEMM_OPS = ["ig_linear_fwd", "ig_linear_residual_fwd", "ig_linear_dgrad", "ig_linear_wgrad", "ig_attention_fwd", "ig_attention_bwd",
            "ig_convT_fwd", "ig_convT_dgrad", "ig_convT_wgrad", "ig_conv3x3_fwd", "ig_conv3x3_dgrad", "ig_conv3x3_wgrad",
            "ig_patch_embed_fwd"]  # fmt: skip
HBM_OPS = ["ig_normalize_chips", "ig_layernorm_fwd", "ig_layernorm_bwd", "ig_colsum", "ig_bn_relu_fwd", "ig_bn_relu_bwd",
           "ig_classifier_fwd", "ig_classifier_bwd", "ig_classifier_bn_fwd", "ig_classifier_bn_bwd", "ig_ce_loss", "ig_adamw_step"]
GEMM_OPS.append("ig_linear_wgrad_group")
TIMED_OPS = ["ig_linear_fwd", "ig_linear_residual_fwd", "ig_linear_dgrad", "ig_linear_wgrad", "ig_linear_wgrad_group"]


def cpu_baseline(target_s: float = 10.0) -> dict:
    """The oracle's reference-semantics path on the host cores (fp32, Prithvi-100M T=1, B=4): kind 'port'.
    (i) BASELINE.json configs[0]: forward of 4 chips, median of >= 10 runs after 2 warm-ups;
    (ii) the training step (fwd + CE + bwd + AdamW) for a bounded sample."""
    import statistics

    from instageo_amd.config import PRESETS
    from oracle import prithvi_oracle as O

    host_cores = os.cpu_count() or 1
    # a batch of 4 chips does not scale to 100+ threads (MKL/oneDNN oversubscription).  Thread sweep of the configs[0] forward on the bench
    # box (256 host cores, round 4): 8 threads 18.1 chips/s, 16 threads 26.2, 32 threads 14.3, 64 threads 8.5, all cores < 0.27 --
    # so the reported figures use 16 intra-op threads (the best setting), and the sweep is reported beside them
    threads = max(1, min(16, host_cores))
    torch.set_num_threads(threads)
    cfg = O.make_config("prithvi_eo_v1_100", 1, 2)
    sd = O.make_state_dict(cfg, seed=1042)
    B = 4
    g = torch.Generator().manual_seed(1042)
    img = torch.randn(B, 6, 1, 224, 224, generator=g)
    lab = torch.randint(0, 2, (B, 224, 224), generator=g)
    with torch.no_grad():
        for _ in range(2):
            O.prithvi_seg_forward(cfg, sd, img, training=False)
        ts = []
        t_all = time.time()
        while len(ts) < 10 or (time.time() - t_all < 0.4 * target_s and len(ts) < 40):
            t0 = time.time()
            O.prithvi_seg_forward(cfg, sd, img, training=False)
            ts.append(time.time() - t0)
    fwd = {"value": round(B / statistics.median(ts), 3), "unit": "chips/s", "runs": len(ts), "median_s": round(statistics.median(ts), 4),
           "workload": "BASELINE.json configs[0]: PrithviSeg forward, 4 random 224x224x6 T=1 chips, fp32"}  # fmt: skip
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k and not k.endswith("pos_embed")]
    params = {k: sd[k].clone().requires_grad_(True) for k in names}
    opt = torch.optim.AdamW(list(params.values()), lr=1e-4, weight_decay=1e-2)
    full = dict(sd)
    cw = torch.tensor([float(w) for w in PRESETS["sen1floods11"]["train"]["class_weights"]])

    def step():
        full.update(params)
        loss = O.seg_loss(O.prithvi_seg_forward(cfg, full, img, training=True), lab, cw, -1)
        opt.zero_grad()
        loss.backward()
        opt.step()

    step()  # warm-up
    t0 = time.time()
    n = 0
    while n < 2 or time.time() - t0 < 0.6 * target_s:
        step()
        n += 1
    dt = time.time() - t0
    out = {"value": round(B * n / dt, 3), "unit": "chips/s", "cores": threads, "host_cores": host_cores, "kind": "port",
           "sample": f"{n} train steps (fwd+CE+bwd+AdamW) of batch {B}, Prithvi-100M T=1, fp32 CPU oracle, {dt:.1f} s; {threads} of "
                     f"{host_cores} host cores (torch intra-op threads)",
           "forward_configs0": fwd}  # fmt: skip
    if host_cores > threads:
        # BASELINE.md section 4 asks for all physical cores with the count stated: the configs[0] forward (B = 4) with EVERY host core as
        # an intra-op thread, reported BESIDE the 32-thread figures.  At batch 4 the all-core run is pathologically slow (256 threads
        # oversubscribe 4 chips: a train step took ~150 s on the bench box), so it runs in a child process under a 45 s watchdog and
        # covers the forward only; a timeout is reported as such.
        import subprocess

        code = ("import sys, time, torch; sys.path[:0] = %r; from oracle import prithvi_oracle as O; torch.set_num_threads(%d);"
                "cfg = O.make_config('prithvi_eo_v1_100', 1, 2); sd = O.make_state_dict(cfg, seed=1042);"
                "img = torch.randn(4, 6, 1, 224, 224, generator=torch.Generator().manual_seed(1042));\n"
                "with torch.no_grad():\n"
                "    O.prithvi_seg_forward(cfg, sd, img, training=False); t0 = time.time(); n = 0\n"
                "    while n < 2 or time.time() - t0 < 5: O.prithvi_seg_forward(cfg, sd, img, training=False); n += 1\n"
                "print('ALLCORE', 4 * n / (time.time() - t0), n)") % ([ROOT, os.path.join(ROOT, "instageo-e2e-geospatial-ml_amd")], 987654321)
        code = code.replace("987654321", "%d")
        def probe(nthreads: int, limit: int):
            try:
                r = subprocess.run([sys.executable, "-c", code % nthreads], capture_output=True, text=True, timeout=limit,
                                   env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
                tok = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("ALLCORE")]
                if tok:
                    return {"forward_configs0_chips_per_s": round(float(tok[0][1]), 3), "runs": int(tok[0][2]), "cores": nthreads}
                return {"forward_configs0_chips_per_s": None, "cores": nthreads, "error": (r.stderr or "no output")[-200:]}
            except subprocess.TimeoutExpired:
                return {"forward_configs0_chips_per_s": None, "cores": nthreads, "timeout_s": limit,
                        "upper_bound_chips_per_s": round(12.0 / limit, 3),
                        "note": f"forward of 4 chips did not finish 3 passes in {limit} s"}

        out["all_cores"] = probe(host_cores, 45)
        if host_cores >= 128:
            out["threads_64"] = probe(64, 20)  # the trend between the best setting and the all-core one
        if host_cores >= 64:
            out["threads_32"] = probe(32, 25)
        if host_cores >= 16:
            out["threads_8"] = probe(8, 30)
        torch.set_num_threads(threads)
    return out


def dry_run(args) -> None:
    """The contract's plumbing without the HIP library: rendezvous from the torchrun environment, W + K "steps" (a 32 MiB
    gradient-bucket-sized all-reduce each, the step's one exchange), barrier-bracketed max-over-ranks timing, the JSON line."""
    from instageo_amd import distributed as D

    rank, local_rank, world = D.init_from_env()
    use_cuda = torch.cuda.is_available() and os.environ.get("IG_DIST_BACKEND", "nccl") != "gloo"
    dev = torch.device("cuda", local_rank) if use_cuda else torch.device("cpu")
    bucket = torch.ones(8 << 20, dtype=torch.float32, device=dev)

    def barrier() -> None:
        if world > 1:
            dist.barrier()
        if use_cuda:
            torch.cuda.synchronize()

    def step() -> None:
        if world > 1:
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
            bucket.div_(world)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    if rank == 0:
        print(json.dumps({
            "metric": "dry run: launch / rendezvous / timing plumbing only (no chips processed)", "value": 0.0, "unit": "chips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / max(1, args.steps), 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "none", "dry_run": True,
            "config": {"workload": "dry run", "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "dist": {"ranks": world, "backend": dist.get_backend() if world > 1 else None,
                     "bucket_checksum": float(bucket[0].item())}}))  # fmt: skip
    if world > 1:
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=432, help="chips per GPU per step (432 x 197 tokens = 333 row tiles of 256: the N = 768 "
                    "GEMMs fill 4 rounds of 256 CUs to 98 %%, attention 20.25 rounds; the per-step fixed costs -- AdamW, folds, tile-round tails -- "
                    "amortise: +4 %% over 216, the default of rounds 4-6 until the last build; 108 = one round was the default of rounds 1-3)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3"])
    ap.add_argument("--model", default="prithvi_eo_v1_100", help="variant (other BASELINE configs: prithvi_eo_v2_300)")
    ap.add_argument("--temporal", type=int, default=1, help="T: 1 = configs[1] (default), 3 = configs[2] multi-temporal crop")
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-leg", action="store_true", help="skip the bf16x3 (1e-3-parity) leg")
    ap.add_argument("--no-tile", action="store_true", help="skip the configs[3] sliding-window leg")
    ap.add_argument("--no-yaml-legs", action="store_true", help="N = 1: skip the legs at the reference YAMLs' batches (16; T = 3: 8) and the T = 3 / batch 72 leg")
    ap.add_argument("--no-t3-leg", action="store_true", help="N > 1: skip the T = 3 / 13-class leg (the shape the scaling target is stated on)")
    ap.add_argument("--t3-batch", type=int, default=72, help="per-GPU batch of the T = 3 leg (72 x 589 tokens = 166 row tiles of 256, the same fill "
                    "of the tile rounds as the default T = 1 batch: +5.7 %% chips/s over 36 on the same box)")
    ap.add_argument("--tile-size", type=int, default=10980)
    ap.add_argument("--event-stride", type=int, default=7, help="bracket every n-th launch of the timed entry points with HIP events")
    ap.add_argument("--no-profile", action="store_true", help="skip per-launch HIP events (roofline objects become null)")
    ap.add_argument("--graph", action="store_true", help="replay the train step from one captured hipGraph (N=1 only; implies --no-profile)")
    ap.add_argument("--detail-file", default=None, help="where rank 0 writes the per-kernel tables (default profiles/bench_detail_*.json)")
    ap.add_argument("--dry-run", action="store_true", help="launch / rendezvous / timing / JSON plumbing only (no HIP library needed)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Self-launch: one rank per GPU through torch.distributed.run, as a CHILD process and before this process has touched
        # the GPU (never exec from a process that has initialised HIP).  127.0.0.1 rendezvous: the hostname may not resolve.
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]  # fmt: skip
        sys.exit(subprocess.run(cmd, env=env).returncode)
    if args.dry_run:
        return dry_run(args)

    from instageo_amd import distributed as D
    from instageo_amd import ops
    from instageo_amd.config import PRESETS
    from instageo_amd.segmentation import PrithviSegmentationModule

    rank, local_rank, world = D.init_from_env()
    dp = D.dp_active()  # more than one rank -- or ONE rank under IG_DIST_FORCE=1 (pre-flight of the RCCL path on a one-GPU box)
    if world != args.gpus:  # every rank: a bench line whose n_gpus differs from the ranks that ran is not a measurement
        print(f"bench.py: --gpus {args.gpus} but {world} rank(s) joined the rendezvous (rank {rank})", file=sys.stderr)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(2)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    torch.manual_seed(1042 + rank)

    crop_w = PRESETS["multitemporal_crop_classification"]["train"]["class_weights"]  # multitemporal_crop_classification.yaml:15-30
    mean = torch.tensor(MEAN, device=dev)
    std = torch.tensor(STD, device=dev)
    nb = 4  # resident synthetic batches (raw int16 HLS domain), cycled

    def make_workload(B: int, T: int, NCLS: int) -> dict:
        """Resident synthetic batches of one configuration (per rank: its own shard of the global batch)."""
        g = torch.Generator(device=dev).manual_seed(1042 + rank)
        raws = [torch.randint(0, 10000, (B, 6 * T, 224, 224), generator=g, device=dev, dtype=torch.int16) for _ in range(nb)]
        labels = []
        for _ in range(nb):
            y = torch.randint(0, NCLS, (B, 224, 224), generator=g, device=dev)
            y[torch.rand((B, 224, 224), generator=g, device=dev) < 0.05] = -1
            labels.append(y)
        return {"B": B, "T": T, "NCLS": NCLS, "raws": raws, "labels": labels,
                "cw": [1, 3] if NCLS == 2 else (crop_w if NCLS == len(crop_w) else [1.0] * NCLS),
                "xbuf": torch.empty((B, 6, T, 224, 224), dtype=torch.float32, device=dev)}

    B = args.batch
    T, NCLS = args.temporal, args.classes
    main_wl = make_workload(B, T, NCLS)
    raws, labels, xbuf = main_wl["raws"], main_wl["labels"], main_wl["xbuf"]

    def barrier() -> None:
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        if dp:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    def run_mode(precision: str, profile: bool, graph: bool, wl: dict = None):
        """Train + inference (+ encoder-forward) legs of one precision mode on workload ``wl``; returns a dict of raw measurements."""
        wl = wl or main_wl
        B, T, NCLS, raws, labels, xbuf, cw = wl["B"], wl["T"], wl["NCLS"], wl["raws"], wl["labels"], wl["xbuf"], wl["cw"]
        mod = PrithviSegmentationModule(image_size=224, learning_rate=1e-4, freeze_backbone=False, load_pretrained_weights=False,
                                        num_classes=NCLS, temporal_step=T, class_weights=cw, ignore_index=-1, weight_decay=0.01,
                                        scheduler=False, model_name=args.model, precision=precision, device=dev)  # fmt: skip
        sync = D.attach_data_parallel(mod)
        stats = torch.zeros(2, dtype=torch.float64, device=dev)
        graphed = None
        if graph and world == 1:
            ops.normalize_chips(raws[0], mean, std, T, 1e-4, out=xbuf)
            graphed = mod.make_graphed_train_step(xbuf, labels[0])

        def train_step(i: int) -> None:
            ops.normalize_chips(raws[i % nb], mean, std, T, 1e-4, out=xbuf)
            if graphed is not None:
                stats.copy_(graphed(xbuf, labels[i % nb]))
            else:
                mod.fused_train_step(xbuf, labels[i % nb], stats=stats)

        for i in range(args.warmup):
            train_step(i)
        barrier()
        if profile:
            # Only the dominant (linear GEMM) entry points carry events inside the timed region, and only every 7th launch of
            # each: an event pair costs a few microseconds of queue drain, which at ~145 GEMM launches per step was 8 % of the
            # step.  7 is coprime with the per-block launch pattern (qkv/proj/fc1/fc2), so the sample keeps the shape mix.
            ops.profile_begin(TIMED_OPS, stride=args.event_stride)
        t0 = time.perf_counter()
        for i in range(args.steps):
            train_step(i)
        barrier()
        dt_own = time.perf_counter() - t0
        dt = max_over_ranks(dt_own)
        per_rank = None
        if dp and dist.is_initialized():  # every rank's own clock around the same K steps (the barrier makes them nearly equal: a straggler shows)
            mine = torch.tensor([dt_own], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [round(B * args.steps / a.item(), 1) for a in allr]
        res = {"mod": mod, "dt": dt, "per_rank_chips_per_s": per_rank, "prof": ops.profile_end() if profile else None, "prof_all": None, "graphed": graphed is not None,
               "loss": (stats[0] / stats[1]).item(), "buckets": None}
        res["dp_mode"] = None if sync is None else ("zero1" if isinstance(sync, D.ShardedGradSync) else "allreduce")
        if isinstance(sync, D.GradSync):  # per-bucket all-reduce time of one extra, instrumented step (outside the timed region)
            sync.time_buckets = True
            train_step(0)
            torch.cuda.synchronize()
            res["buckets"] = [{"mbytes": round((hi - lo) * 4 / 2**20, 1), "ms": round(ms, 3)} for (lo, hi), ms in sync.bucket_times()]
            sync.time_buckets = False
        elif sync is not None:
            # reduce-scatter + all-gather around the sharded optimizer: bucket sizes, and the EXPOSED communication of one extra,
            # instrumented step (events around every wait on the compute stream: the time it stands still for the exchange)
            res["buckets"] = [{"mbytes": round((hi - lo) * 4 / 2**20, 1), "ms": 0.0} for (lo, hi, _) in sync.plan]
            try:
                sync.time_exposed = True
                train_step(0)
                res["exposed"] = sync.exposed_ms()
            except Exception:  # noqa: BLE001  (diagnostics only)
                res["exposed"] = None
            sync.time_exposed = False
        if profile:  # every MFMA / HBM entry point, in a separate untimed pass of 3 steps
            # (the optimizer as ONE launch after the backward pass here: launched per gradient range on the side stream, as the timed
            # region does, its event pairs would include the wait for the launch stream)
            mod._early_ok = False
            ops.profile_begin(GEMM_OPS + HBM_OPS)
            for i in range(3):
                train_step(i)
            res["prof_all"] = ops.profile_end()
            mod._early_ok = True
        # inference leg: K0 + forward + argmax(int8)  (chip_inference loop, infer_utils.py:93-101)
        mod.net.eval()
        pred = torch.empty((B, 224, 224), dtype=torch.int8, device=dev)
        eng = mod.net.engine

        def infer_step(i: int) -> None:
            ops.normalize_chips(raws[i % nb], mean, std, T, 1e-4, out=xbuf)
            ops.argmax_i8(eng.forward(xbuf, training=False, save=False), pred)

        with torch.no_grad():
            for i in range(max(2, args.warmup // 2)):
                infer_step(i)
            barrier()
            t1 = time.perf_counter()
            for i in range(args.steps):
                infer_step(i)
            barrier()
            res["dti"] = max_over_ranks(time.perf_counter() - t1)
            # encoder forward alone: HIP events on the launch stream around K passes
            for _ in range(2):
                eng.encoder_forward(xbuf, save=False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                eng.encoder_forward(xbuf, save=False)
            e1.record()
            torch.cuda.synchronize()
            res["enc_ms"] = e0.elapsed_time(e1) / args.steps
        return res

    def dp_preflight() -> dict:
        """Before anything is timed on a data-parallel run: every rank agrees on the world size, and the DEFERRED all-gather of the bf16 operand
        copy (IG_DP_DEFER=1, the default: waited for Block by Block by the next forward) leaves exactly the operand copy and fp32 masters that
        the blocking form leaves -- two modules from the same seed take two steps on the same batch, one per form, and a checksum of
        ``store.shadow`` / ``store.flat`` is compared on every rank.  A difference switches this process to the blocking form (the variable is
        read per step; no re-exec) and is reported on the JSON line."""
        t = torch.tensor([world, dist.get_world_size(), 1], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ranks_seen = int(t[2].item())
        if ranks_seen != world or int(t[0].item()) != world * world or int(t[1].item()) != world * world:
            raise SystemExit(f"bench.py pre-flight: {ranks_seen} ranks answered, WORLD_SIZE={world}, get_world_size()={dist.get_world_size()}")
        info = {"ranks_seen": ranks_seen, "defer": os.environ.get("IG_DP_DEFER", "1") != "0", "defer_fallback": False}
        if os.environ.get("IG_DP_MODE", "zero1") != "zero1" or not info["defer"]:
            return info
        sums = []
        for form in ("1", "0"):
            os.environ["IG_DP_DEFER"] = form
            torch.manual_seed(4242)
            m = PrithviSegmentationModule(image_size=224, learning_rate=1e-4, freeze_backbone=False, load_pretrained_weights=False, num_classes=NCLS,
                                          temporal_step=T, class_weights=main_wl["cw"], ignore_index=-1, weight_decay=0.01, scheduler=False,
                                          model_name=args.model, precision=args.precision, device=dev)  # fmt: skip
            sy = D.attach_data_parallel(m)
            pb = min(B, 8)
            ops.normalize_chips(raws[0][:pb], mean, std, T, 1e-4, out=xbuf[:pb])
            for _ in range(2):
                m.fused_train_step(xbuf[:pb], labels[0][:pb])
            if hasattr(sy, "wait_params"):
                sy.wait_params()
            torch.cuda.synchronize()
            st = m.net.store
            sums.append((st.shadow.hi.view(torch.int16).to(torch.int64).sum().item(),))  # the bf16 operand copy every kernel of the next forward reads
            del m, sy
        os.environ["IG_DP_DEFER"] = "1"
        bad = torch.tensor([int(sums[0][0] != sums[1][0])], dtype=torch.int64, device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        info["shadow_checksum"] = sums[0][0]
        if bad.item():
            os.environ["IG_DP_DEFER"] = "0"  # every rank takes the same decision (the MAX above)
            info["defer"], info["defer_fallback"] = False, True
            if rank == 0:
                print("bench.py pre-flight: deferred all-gather != blocking all-gather on the operand copy; falling back to IG_DP_DEFER=0", file=sys.stderr)
        torch.cuda.empty_cache()
        return info

    preflight = dp_preflight() if dp and dist.is_initialized() else None
    main_res = run_mode(args.precision, not args.no_profile and not args.graph, args.graph)
    hbm_peak_gib = round(torch.cuda.max_memory_allocated() / 2**30, 2) if torch.cuda.is_available() else None  # the main legs (train + inference)
    mod = main_res["mod"]
    cfgm = mod.net.cfg
    fpc, fpc_enc = flop_per_chip_fwd(cfgm.embed_dim, cfgm.depth, T, NCLS)

    # PCIe-inclusive view (never ``value``): one batch of raw int16 chips + float labels from PINNED host memory, (i) copied
    # serially in front of every step, (ii) copied on a side stream into the other half of a double buffer while the step runs.
    host_leg = None
    if world == 1 and not args.graph:
        h_raw = torch.empty(raws[0].shape, dtype=raws[0].dtype).pin_memory()
        h_lab = torch.empty(labels[0].shape, dtype=labels[0].dtype).pin_memory()
        h_raw.copy_(raws[0]), h_lab.copy_(labels[0])
        d_raw = [torch.empty_like(raws[0]) for _ in range(2)]
        d_lab = [torch.empty_like(labels[0]) for _ in range(2)]
        stats_h = torch.zeros(2, dtype=torch.float64, device=dev)
        mod.net.train()

        def step_on(j: int) -> None:
            ops.normalize_chips(d_raw[j], mean, std, T, 1e-4, out=xbuf)
            mod.fused_train_step(xbuf, d_lab[j], stats=stats_h)

        nbytes = h_raw.numel() * h_raw.element_size() + h_lab.numel() * h_lab.element_size()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            d_raw[0].copy_(h_raw, non_blocking=True), d_lab[0].copy_(h_lab, non_blocking=True)
        e1.record()
        torch.cuda.synchronize()
        copy_ms = e0.elapsed_time(e1) / 5
        k = max(4, args.steps // 2)
        for _ in range(2):
            step_on(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            d_raw[0].copy_(h_raw, non_blocking=True), d_lab[0].copy_(h_lab, non_blocking=True)
            step_on(0)
        torch.cuda.synchronize()
        serial = B * k / (time.perf_counter() - t0)
        side = torch.cuda.Stream()
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        freed = [torch.cuda.Event(), torch.cuda.Event()]
        for j in range(2):
            freed[j].record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k + 1):
            j = i & 1
            if i < k:  # prefetch batch i into buffer j on the side stream once the step that last used it is done
                with torch.cuda.stream(side):
                    side.wait_event(freed[j])
                    d_raw[j].copy_(h_raw, non_blocking=True), d_lab[j].copy_(h_lab, non_blocking=True)
                    ready[j].record(side)
            if i > 0:
                torch.cuda.current_stream().wait_event(ready[j ^ 1])
                step_on(j ^ 1)
                freed[j ^ 1].record()
        torch.cuda.synchronize()
        overlapped = B * k / (time.perf_counter() - t0)
        host_leg = {"batch_mbytes": round(nbytes / 2**20, 1), "h2d_ms": round(copy_ms, 3), "h2d_GBps": round(nbytes / copy_ms / 1e6, 1),
                    "train_chips_per_s_serial_copy": round(serial, 1), "train_chips_per_s_overlapped_copy": round(overlapped, 1),
                    "note": "pinned host memory -> HBM over PCIe; int16 chips + f32 labels of one batch; never the headline value"}

    tile_leg = None
    if not args.no_tile and (T, args.model) == (1, "prithvi_eo_v1_100"):
        # BASELINE.json configs[3]: one resident 6 x S x S int16 tile -> (S // 224)^2 windows, gather + normalise + forward + argmax.
        # N > 1: every rank holds the tile, takes a contiguous block of the window list (shard_range: 7 x 300 + 301 at N = 8) with no
        # data-path collective, and the int8 maps are gathered on rank 0 over RCCL at the end.
        from instageo_amd.infer_utils import sliding_window_inference

        S = args.tile_size
        gt = torch.Generator(device=dev).manual_seed(7)
        tile = torch.randint(0, 10000, (6, S, S), generator=gt, device=dev, dtype=torch.int16)
        # warm-up = one untimed pass over the same tile: the engine allocates one workspace per batch size (full batches and the
        # ragged last one), which the first pass of a process pays once (13977 vs 15155 windows/s); a tile service processes many tiles
        sliding_window_inference(tile, mod, MEAN, STD, 1, 224, 224, batch_size=B, constant_multiplier=1e-4)
        barrier()
        t0 = time.perf_counter()
        local_maps, origins = sliding_window_inference(tile, mod, MEAN, STD, 1, 224, 224, batch_size=B, constant_multiplier=1e-4, gather=False)
        torch.cuda.synchronize()
        dt_local = time.perf_counter() - t0  # this rank's windows, no collective
        n_local = int(local_maps.shape[0])
        if dp:
            counts = [D.shard_range(len(origins), r, world)[1] - D.shard_range(len(origins), r, world)[0] for r in range(world)]
            maps = D.gather_class_maps(local_maps, counts, dst=0)
        else:
            maps = local_maps
        barrier()
        dtt = max_over_ranks(time.perf_counter() - t0)  # whole tile including the final gather
        per_gpu = 1.0 / max_over_ranks(dt_local / max(1, n_local))  # slowest rank's windows/s
        tile_leg = {"workload": f"BASELINE.json configs[3]: sliding-window chip_inference over a resident 6x{S}x{S} int16 tile, "
                                f"{len(origins)} windows of 224 (stride 224) sharded over {world} rank(s), batch {B}, window gather + normalise "
                                f"included; second pass over the tile (workspaces allocated); value = whole tile incl. the final gather of the int8 maps",
                    "value": round(len(origins) / dtt, 1), "unit": "windows/s", "seconds": round(dtt, 4), "windows": len(origins),
                    "per_gpu_windows_per_s": round(per_gpu, 1), "windows_per_rank": n_local,
                    "class_histogram": torch.bincount(maps.flatten().long() + 1, minlength=NCLS + 1)[1:].tolist() if maps is not None else None}  # fmt: skip
        del tile, maps, local_maps

    parity = None
    if args.precision == "bf16" and not args.no_parity_leg and not args.graph:
        del mod
        main_mod_freed = main_res.pop("mod")
        del main_mod_freed
        torch.cuda.empty_cache()
        parity = run_mode("bf16x3", False, False)
        parity.pop("mod")
        torch.cuda.empty_cache()

    t3_leg = None
    if dp and (T, NCLS, args.model) == (1, 2, "prithvi_eo_v1_100") and not args.no_t3_leg and not args.graph:
        # the shape north_star's >= 7x scaling target is stated on (BASELINE configs[2]: 224x224x6x3 chips, 13 classes), beside the
        # default line: same data-parallel step, per-GPU batch --t3-batch
        main_res.pop("mod", None)
        mod = None
        torch.cuda.empty_cache()
        try:  # an extra leg must never cost the headline line (every rank takes the same branch: the failure modes are symmetric)
            wl3 = make_workload(args.t3_batch, 3, 13)
            r3 = run_mode(args.precision, False, False, wl3)
            r3.pop("mod")
            t3_leg = {"workload": "prithvi_eo_v1_100 T=3 13 classes (BASELINE configs[2] shape), synthetic int16 chips", "per_gpu_batch": args.t3_batch,
                      "value": round(world * args.t3_batch * args.steps / r3["dt"], 2), "unit": "chips/s", "ms_per_step": round(1e3 * r3["dt"] / args.steps, 3),
                      "inference_chips_per_s": round(world * args.t3_batch * args.steps / r3["dti"], 1), "exposed_comm_ms": r3.get("exposed")}
            del wl3
        except Exception as e:  # noqa: BLE001
            t3_leg = {"error": f"{e.__class__.__name__}: {e}"[:300], "value": None, "ms_per_step": None}
        torch.cuda.empty_cache()

    # N = 1, headline configuration: the reference YAMLs' own per-GPU batches (configs/sen1floods11.yaml:13 batch_size 16;
    # multitemporal_crop_classification.yaml:14 batch_size 8, T = 3 / 13 classes) and configs[2]'s shape at the batch that fills the
    # tile rounds (72) as extra, short legs -- SURVEY 8(d) cfg 2: "bs 16 and the largest B that fits" -- so that the small-batch numbers
    # are driver-observed
    yaml_legs = {}
    if not dp and (T, NCLS, args.model) == (1, 2, "prithvi_eo_v1_100") and args.precision == "bf16" and not args.no_yaml_legs and not args.graph:
        main_res.pop("mod", None)
        mod = None
        torch.cuda.empty_cache()
        for key, (b_, t_, c_) in {"yaml_b16": (16, 1, 2), "yaml_t3_b8": (8, 3, 13), "t3_b72": (args.t3_batch, 3, 13)}.items():
            try:  # an extra leg must never cost the headline line
                wl_ = make_workload(b_, t_, c_)
                r_ = run_mode(args.precision, False, False, wl_)
                r_.pop("mod")
                yaml_legs[key] = {"per_gpu_batch": b_, "temporal": t_, "classes": c_, "value": round(b_ * args.steps / r_["dt"], 1), "unit": "chips/s",
                                  "ms_per_step": round(1e3 * r_["dt"] / args.steps, 3), "inference_chips_per_s": round(b_ * args.steps / r_["dti"], 1),
                                  "encoder_fwd_ms": round(r_["enc_ms"], 3)}
                del wl_, r_
            except Exception as e:  # noqa: BLE001
                yaml_legs[key] = {"error": f"{e.__class__.__name__}: {e}"[:300], "value": None}
            torch.cuda.empty_cache()

    if rank != 0:
        if dp:
            dist.destroy_process_group()
        return

    dt, dti = main_res["dt"], main_res["dti"]
    value = world * B * args.steps / dt
    headline = (T, NCLS, args.model) == (1, 2, "prithvi_eo_v1_100")
    enc_tflops = B * fpc_enc / (main_res["enc_ms"] * 1e-3) / 1e12
    cfg_out = {"workload": "BASELINE.json configs[1]: Prithvi-100M fine-tune, Sen1Floods11-shaped synthetic int16 chips (6 bands, T=1, "
                           "224x224, 2 classes, weights [1,3], ignore -1, dropout 0.1)" if headline else
                           f"{args.model} T={T} {NCLS} classes, synthetic int16 chips, dropout 0.1, random init",
               "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
               "launch": "hipGraph" if main_res["graphed"] else "eager",
               "deterministic": os.environ.get("IG_DETERMINISTIC", "1") != "0",  # bit-reproducible reductions (reference: Trainer(deterministic=True))
               "final_loss": round(main_res["loss"], 5),
               "whole_step_mfma_frac": round(value / world * 3 * fpc / (PEAK_BF16_TFLOPS * 1e12), 4),
               "encoder_fwd_ms": round(main_res["enc_ms"], 3), "encoder_fwd_tflops": round(enc_tflops, 1),
               "encoder_fwd_mfma_frac": round(enc_tflops / PEAK_BF16_TFLOPS, 4),
               "inference_chips_per_s": round(world * B * args.steps / dti, 1),
               "inference_mfma_frac": round(B * args.steps / dti * fpc / (PEAK_BF16_TFLOPS * 1e12), 4)}  # fmt: skip
    detail = {"gflop_per_chip_fwd": round(fpc / 1e9, 2), "gflop_per_chip_encoder_fwd": round(fpc_enc / 1e9, 2),
              "inference_ms_per_step": round(1e3 * dti / args.steps, 3), "optimizer": "AdamW lr 1e-4 wd 1e-2",
              "hbm_peak_gib_torch_allocator": hbm_peak_gib}  # (the library's own scratch -- slabs, packed weights -- is outside the allocator: < 1 GiB)
    if parity is not None:
        pv = world * B * args.steps / parity["dt"]
        pi = world * B * args.steps / parity["dti"]
        cfg_out["parity_mode_chips_per_s"] = round(pv, 1)
        cfg_out["parity_mode_inference_chips_per_s"] = round(pi, 1)
        cfg_out["parity_mode_frac_of_x3_ceiling"] = round(pv / world * 3 * fpc / (PEAK_BF16_TFLOPS / 3 * 1e12), 4)
        detail["parity_mode"] = {
            "dtype": "bf16x3", "value": round(pv, 2), "unit": "chips/s", "ms_per_step": round(1e3 * parity["dt"] / args.steps, 3),
            "inference": {"value": round(pi, 2), "unit": "chips/s", "ms_per_step": round(1e3 * parity["dti"] / args.steps, 3)},
            "encoder_fwd_ms": round(parity["enc_ms"], 3), "final_loss": round(parity["loss"], 5),
            "note": "same workload, batch and step; split-bf16 operands (hi*hi + hi*lo + lo*hi, 3 MFMAs per product: ceiling = "
                    "peak / 3); this mode meets the north-star 1e-3 logits / mIoU tolerance (tests/test_gpu_model.py)"}  # fmt: skip
    for key, leg in yaml_legs.items():
        cfg_out[key + "_chips_per_s"] = leg["value"]
    if yaml_legs:
        detail["yaml_batch_legs"] = yaml_legs
    if tile_leg is not None:
        cfg_out["tile_windows_per_s"] = tile_leg["value"]
        cfg_out["tile_windows_per_s_per_gpu"] = tile_leg["per_gpu_windows_per_s"]
        detail["tile_inference"] = tile_leg
    if host_leg is not None:
        cfg_out["train_chips_per_s_pcie_overlapped"] = host_leg["train_chips_per_s_overlapped_copy"]
        detail["host_input"] = host_leg
    out = {
        "metric": f"HLS chips/sec (train fwd+bwd+AdamW), {args.model} 224x224x6 T={T}",
        "value": round(value, 2),
        "unit": "chips/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.precision,
        "data": "synthetic",
        "config": cfg_out,
    }
    prof, prof_all = main_res["prof"], main_res["prof_all"]
    if prof is not None:
        def table(p, per_step):
            return {name: {"launches_per_step": round(r["calls"] / per_step, 2), "timed_launches": r["n"], "avg_us": round(1e3 * r["ms"] / r["n"], 2),
                           "ms_per_step": round(r["ms"] / r["n"] * r["calls"] / per_step, 3),
                           "gflop_per_launch": round(r["work"] / r["n"] / 1e9, 2),
                           "achieved_tflops": round(r["work"] / (r["ms"] * 1e-3) / 1e12, 1),
                           "frac": round(r["work"] / (r["ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)}
                    for name, r in p.items() if r["n"]}  # fmt: skip

        timed_k = table(prof["kernels"], args.steps)
        all_k = table({k: v for k, v in prof_all["kernels"].items() if v["op"] in GEMM_OPS}, 3)
        # HBM-bound entry points: algorithmic bytes / HIP-event time against the 8 TB/s HBM3E peak
        detail["hbm_ops"] = {name: {"launches": n, "avg_us": round(1e3 * ms / n, 2), "total_ms": round(ms, 2),
                                    "achieved_GBps": round(work / (ms * 1e-3) / 1e9, 1),
                                    "frac_of_peak": round(work / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 3)}
                             for name, (n, ms, work) in prof_all["ops"].items() if n and name in HBM_OPS}  # fmt: skip
        dom = max(timed_k, key=lambda k: timed_k[k]["ms_per_step"])
        d = timed_k[dom]
        out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": d["achieved_tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                           "frac": d["frac"], "traffic": None, "launches_per_step": d["launches_per_step"], "avg_launch_us": d["avg_us"],
                           "ms_per_step": d["ms_per_step"], "gflop_per_launch": d["gflop_per_launch"], "timed_launches": d["timed_launches"],
                           "event_stride": args.event_stride}  # fmt: skip
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_bench_b{B}.json")))
        if pmc_files and headline:
            # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
            # (tools/pmc_bench.sh); FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md
            pmc = json.load(open(pmc_files[-1]))
            # (rocprofv3 prints defaulted template arguments the library's launch notes leave out: gemm4_kernel<0,0,false> is <0, 0, false, 1>)
            ent = next((v for k, v in pmc.items() if k.replace(" ", "") in (dom, dom[:-1] + ",1>")), None)
            if ent:
                out["roofline"]["traffic"] = round((2 * ent["fetch_kb_raw"] + ent["write_kb"]) * 1024)
                out["roofline"]["traffic_src"] = os.path.relpath(pmc_files[-1], ROOT)
        detail["roofline_note"] = ("dominant kernel of the timed region by launches/step x average launch; HIP events on the launch stream, "
                                   f"every {args.event_stride}-th launch of the linear-GEMM entry points")
        detail["roofline_timed_region"] = timed_k
        detail["roofline_kernels"] = all_k
    if dp:
        bk = main_res["buckets"] or []
        out["dist"] = {"ranks": world, "backend": dist.get_backend(), "mode": main_res.get("dp_mode"), "reserved_cus": ops.reserved_cus(), "buckets": len(bk),
                       "per_rank_chips_per_s": main_res.get("per_rank_chips_per_s"), "preflight": preflight,
                       "allreduce_mbytes": round(sum(b["mbytes"] for b in bk), 1), "allreduce_ms_serial": round(sum(b["ms"] for b in bk), 3)}
        ex = main_res.get("exposed")
        if ex is not None:  # zero1: time the compute stream stood still for the reduce-scatters / all-gathers of one instrumented step
            out["dist"].update({"rs_exposed_ms": round(ex["rs"], 3), "ag_exposed_ms": round(ex["ag"], 3), "tail_exposed_ms": round(ex["tail"], 3),
                                "gather": "shadow"})
        if t3_leg is not None:
            out["dist"]["t3_c13_chips_per_s"] = t3_leg["value"]
            out["dist"]["t3_c13_ms_per_step"] = t3_leg["ms_per_step"]
            detail["dp_t3_leg"] = t3_leg
        detail["allreduce_buckets"] = bk
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline()
        detail["cpu_baseline"] = cb
        out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                               "sample": f"oracle train steps (fwd+CE+bwd+AdamW), batch 4, Prithvi-100M T=1 fp32, {cb['cores']} of {cb['host_cores']} host cores "
                                         "(best of a thread sweep)",
                               "host_cores": cb["host_cores"], "forward_configs0_chips_per_s": cb["forward_configs0"]["value"],
                               "forward_configs0_all_cores": (cb.get("all_cores") or {}).get("forward_configs0_chips_per_s"),
                               "all_cores_upper_bound": (cb.get("all_cores") or {}).get("upper_bound_chips_per_s")}  # (thread sweep: detail file)
    path = args.detail_file or os.path.join(ROOT, "profiles", f"bench_detail_n{world}_b{B}_{args.model}_t{T}.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"line": out, "detail": detail}, f, indent=1)
        out["detail"] = os.path.relpath(path, ROOT)
    except OSError as e:  # read-only checkout: the line still carries the headline numbers
        out["detail"] = f"not written ({e.__class__.__name__})"
    line = json.dumps(out)
    if len(line) > 1950:  # the driver keeps a 2000-character tail: never let the line outgrow it.  Optional fields go first ...
        for k in ("forward_configs0_thread_sweep", "all_cores_upper_bound", "forward_configs0_all_cores", "forward_configs0_chips_per_s"):
            out.get("cpu_baseline", {}).pop(k, None)
        out["config"].pop("train_chips_per_s_pcie_overlapped", None)
        out["config"].pop("tile_windows_per_s_per_gpu", None)
        line = json.dumps(out)
    if len(line) > 1950:  # ... then the workload text; the contract's keys (cpu_baseline.sample among them) stay
        out["config"]["workload"] = out["config"]["workload"][:60]
        line = json.dumps(out)
    if len(line) > 1950:
        out.get("cpu_baseline", {})["sample"] = out.get("cpu_baseline", {}).get("sample", "")[:60]
        line = json.dumps(out)
    print(line)
    if dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
