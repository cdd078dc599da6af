#!/usr/bin/env python3
"""Headline benchmark: Prithvi-100M segmentation training step (fwd + CE + bwd + AdamW) in chips/s.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (for N>1 launched by torch.distributed.run, one
rank per GPU over RCCL).  Prints ONE JSON line on rank 0.  Workload = BASELINE.json configs[1]: Prithvi-100M
fine-tune on Sen1Floods11-shaped synthetic chips (6 bands, T=1, 224x224, 2 classes, class weights [1,3],
ignore_index -1, dropout on), bf16 MFMA, per-GPU batch ``--batch`` (weak scaling).  A step = K0 normalise of a
resident int16 batch -> forward -> loss/metrics -> backward -> (gradient all-reduce) -> AdamW.

Extra objects: ``roofline`` (dominant MFMA kernel: algorithmic FLOPs / HIP-event time per launch over the timed
region), ``roofline_all`` (every profiled entry point), ``inference`` (fwd + argmax chips/s), and at N=1
``cpu_baseline`` (the CPU oracle's training step timed on the host cores).
"""
from __future__ import annotations

import argparse
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
FLOP_PER_CHIP_FWD = 47.85e9  # 100M, T=1, 2 classes


def flop_per_chip_fwd(D: int, L: int, T: int, ncls: int) -> float:
    """blocks 24*N*D^2 + 4*N^2*D, patch embed, 4 head stages, classifier."""
    N = 1 + 196 * T
    enc = L * (24.0 * N * D * D + 4.0 * N * N * D) + 2.0 * (196 * T) * 1536 * D
    head = 0.0
    for i in range(4):
        cin, hin = D * T / 2**i, 14 * 2**i
        cout = cin / 2
        head += 2 * cin * cout * 9 * hin**2 + 2 * cout * cout * 9 * (2 * hin) ** 2
    return enc + head + 2.0 * (D * T / 16) * ncls * 224**2
GEMM_OPS = ["ig_linear_fwd", "ig_linear_residual_fwd", "ig_linear_dgrad", "ig_linear_wgrad", "ig_attention_fwd", "ig_attention_bwd",
            "ig_convT_fwd", "ig_convT_dgrad", "ig_convT_wgrad", "ig_conv3x3_fwd", "ig_conv3x3_dgrad", "ig_conv3x3_wgrad",
            "ig_patch_embed_fwd"]  # fmt: skip
HBM_OPS = ["ig_normalize_chips", "ig_layernorm_fwd", "ig_layernorm_bwd", "ig_colsum", "ig_bn_relu_fwd", "ig_bn_relu_bwd",
           "ig_classifier_fwd", "ig_classifier_bwd", "ig_ce_loss", "ig_adamw_step"]
TIMED_OPS = ["ig_linear_fwd", "ig_linear_residual_fwd", "ig_linear_dgrad", "ig_linear_wgrad"]
KERNEL_OF = {
    "ig_linear_fwd": "gemm2_kernel<PlainLoader,PlainLoader,EpStore,false,false,1,32>",
    "ig_linear_residual_fwd": "gemm2_kernel<PlainLoader,PlainLoader,EpResidual,false,false,1,32>",  # fc2 (K = 4D) runs on gemm5_kernel
    "ig_linear_dgrad": "gemm5_kernel<PlainLoader,PlainLoader,EpGradStore,false,true,1>",  # d_fc2 (with gelu') stays on gemm2_kernel
    "ig_linear_wgrad": "gemm2_kernel<PlainLoader,PlainLoader,EpAtomic,true,true,1,32,2>",
}


def cpu_baseline(target_s: float = 12.0) -> dict:
    """The oracle's reference-semantics training step (fp32, B=4, 100M T=1) on the host cores: kind 'port'."""
    from oracle import prithvi_oracle as O
    from oracle.cases import class_weights_for

    # a batch of 4 chips does not scale to 100+ threads (MKL/oneDNN oversubscription made it 4x slower than 8 threads)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))

    cfg = O.make_config("prithvi_eo_v1_100", 1, 2)
    sd = O.make_state_dict(cfg, seed=1042)
    B = 4
    g = torch.Generator().manual_seed(1042)
    img = torch.randn(B, 6, 1, 224, 224, generator=g)
    lab = torch.randint(0, 2, (B, 224, 224), generator=g)
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k and not k.endswith("pos_embed")]
    params = {k: sd[k].clone().requires_grad_(True) for k in names}
    opt = torch.optim.AdamW(list(params.values()), lr=1e-4, weight_decay=1e-2)
    full = dict(sd)
    cw = class_weights_for(2)

    def step():
        full.update(params)
        loss = O.seg_loss(O.prithvi_seg_forward(cfg, full, img, training=True), lab, cw, -1)
        opt.zero_grad()
        loss.backward()
        opt.step()

    step()  # warm-up
    t0 = time.time()
    n = 0
    while n < 2 or time.time() - t0 < target_s:
        step()
        n += 1
    dt = time.time() - t0
    return {"value": round(B * n / dt, 3), "unit": "chips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} train steps (fwd+CE+bwd+AdamW) of batch {B}, Prithvi-100M T=1, fp32 CPU oracle, {dt:.1f} s"}  # fmt: skip


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=108, help="chips per GPU per step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3"])
    ap.add_argument("--model", default="prithvi_eo_v1_100", help="variant (other BASELINE configs: prithvi_eo_v2_300)")
    ap.add_argument("--temporal", type=int, default=1, help="T: 1 = configs[1] (default), 3 = configs[2] multi-temporal crop")
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-stride", type=int, default=7, help="bracket every n-th launch of the timed entry points with HIP events")
    ap.add_argument("--no-profile", action="store_true", help="skip per-launch HIP events (roofline objects become null)")
    ap.add_argument("--graph", action="store_true", help="replay the train step from one captured hipGraph (N=1 only; implies --no-profile)")
    args = ap.parse_args()

    from instageo_amd import distributed as D
    from instageo_amd import ops
    from instageo_amd.segmentation import PrithviSegmentationModule

    rank, local_rank, world = D.init_from_env()
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    torch.manual_seed(1042 + rank)

    B = args.batch
    T, NCLS = args.temporal, args.classes
    from oracle.cases import CROP_WEIGHTS  # data only (class weights of multitemporal_crop_classification.yaml)

    cw = [1, 3] if NCLS == 2 else (CROP_WEIGHTS if NCLS == 13 else [1.0] * NCLS)
    mod = PrithviSegmentationModule(image_size=224, learning_rate=1e-4, freeze_backbone=False, load_pretrained_weights=False,
                                    num_classes=NCLS, temporal_step=T, class_weights=cw, ignore_index=-1, weight_decay=0.01,
                                    scheduler=False, model_name=args.model, precision=args.precision, device=dev)  # fmt: skip
    D.attach_data_parallel(mod)
    mean = torch.tensor(MEAN, device=dev)
    std = torch.tensor(STD, device=dev)
    nb = 4  # resident synthetic batches (raw int16 HLS domain), cycled
    g = torch.Generator(device=dev).manual_seed(1042 + rank)
    raws = [torch.randint(0, 10000, (B, 6 * T, 224, 224), generator=g, device=dev, dtype=torch.int16) for _ in range(nb)]
    labels = []
    for _ in range(nb):
        y = torch.randint(0, NCLS, (B, 224, 224), generator=g, device=dev)
        y[torch.rand((B, 224, 224), generator=g, device=dev) < 0.05] = -1
        labels.append(y)
    xbuf = torch.empty((B, 6, T, 224, 224), dtype=torch.float32, device=dev)
    stats = torch.zeros(2, dtype=torch.float64, device=dev)

    graphed = None
    if args.graph and world == 1:
        args.no_profile = True
        ops.normalize_chips(raws[0], mean, std, T, 1e-4, out=xbuf)
        graphed = mod.make_graphed_train_step(xbuf, labels[0])

    def train_step(i: int) -> None:
        ops.normalize_chips(raws[i % nb], mean, std, T, 1e-4, out=xbuf)
        if graphed is not None:
            stats.copy_(graphed(xbuf, labels[i % nb]))
        else:
            mod.fused_train_step(xbuf, labels[i % nb], stats=stats)

    def barrier() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        train_step(i)
    barrier()
    if not args.no_profile:
        # Only the dominant (linear GEMM) entry points carry events inside the timed region, and only every 7th launch of
        # each: an event pair costs a few microseconds of queue drain, which at ~145 GEMM launches per step was 8 % of the
        # step.  7 is coprime with the per-block launch pattern (qkv/proj/fc1/fc2), so the sample keeps the shape mix.
        ops.profile_begin(TIMED_OPS, stride=args.event_stride)
    t0 = time.perf_counter()
    for i in range(args.steps):
        train_step(i)
    barrier()
    dt = time.perf_counter() - t0
    prof = ops.profile_end() if not args.no_profile else None
    loss = (stats[0] / stats[1]).item()
    prof_all = None
    if not args.no_profile:  # every MFMA entry point, in a separate untimed pass of 3 steps
        ops.profile_begin(GEMM_OPS + HBM_OPS)
        for i in range(3):
            train_step(i)
        prof_all = ops.profile_end()
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()

    # inference leg: K0 + forward + argmax(int8)  (chip_inference loop, infer_utils.py:93-101)
    mod.net.eval()
    pred = torch.empty((B, 224, 224), dtype=torch.int8, device=dev)

    def infer_step(i: int) -> None:
        ops.normalize_chips(raws[i % nb], mean, std, T, 1e-4, out=xbuf)
        logits = mod.net.engine.forward(xbuf, training=False, save=False)
        ops.argmax_i8(logits, pred)

    with torch.no_grad():
        for i in range(max(2, args.warmup // 2)):
            infer_step(i)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            infer_step(i)
        barrier()
        dti = time.perf_counter() - t1
    ti = torch.tensor([dti], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(ti, op=dist.ReduceOp.MAX)
    dti = ti.item()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = world * B * args.steps / dt
    cfgm = mod.net.cfg
    fpc = flop_per_chip_fwd(cfgm.embed_dim, cfgm.depth, T, NCLS)
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
        "config": {"workload": ("BASELINE.json configs[1]: Prithvi-100M fine-tune, Sen1Floods11-shaped synthetic int16 chips "
                                "(6 bands, T=1, 224x224, 2 classes, class_weights [1,3], ignore_index -1, dropout 0.1), random-init weights")
                   if (T, NCLS, args.model) == (1, 2, "prithvi_eo_v1_100") else
                   f"{args.model} T={T} {NCLS} classes, synthetic int16 chips, dropout 0.1, random-init weights",
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}", "optimizer": "AdamW lr 1e-4 wd 1e-2",
                   "launch": "hipGraph" if graphed is not None else "eager",
                   "final_loss": round(loss, 5)},  # fmt: skip
        "mfma_frac_whole_step": round(value / world * 3 * fpc / (PEAK_BF16_TFLOPS * 1e12), 4),
        "gflop_per_chip_fwd": round(fpc / 1e9, 2),
        "inference": {"value": round(world * B * args.steps / dti, 2), "unit": "chips/s", "ms_per_step": round(1e3 * dti / args.steps, 3),
                      "mfma_frac": round(B * args.steps / dti * fpc / (PEAK_BF16_TFLOPS * 1e12), 4)},  # fmt: skip
    }
    if prof is not None:
        def table(p):
            return {name: {"launches": n, "avg_us": round(1e3 * ms / n, 2), "total_ms": round(ms, 2),
                           "achieved_tflops": round(work / (ms * 1e-3) / 1e12, 1)} for name, (n, ms, work) in p.items() if n}  # fmt: skip

        timed = table(prof)
        allk = table({k: v for k, v in prof_all.items() if k in GEMM_OPS})
        # HBM-bound entry points: algorithmic bytes / HIP-event time against the 8 TB/s HBM3E peak
        out["hbm_ops"] = {name: {"launches": n, "avg_us": round(1e3 * ms / n, 2), "total_ms": round(ms, 2),
                                 "achieved_GBps": round(work / (ms * 1e-3) / 1e9, 1),
                                 "frac_of_peak": round(work / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 3)}
                          for name, (n, ms, work) in prof_all.items() if n and name in HBM_OPS}  # fmt: skip
        dom = max(timed, key=lambda k: timed[k]["total_ms"])
        n, ms, work = prof[dom]
        ach = work / (ms * 1e-3) / 1e12
        out["roofline"] = {"kernel": KERNEL_OF.get(dom, dom), "entry_point": dom, "bound": "mfma", "achieved": round(ach, 1),
                           "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None,
                           "launches": n, "avg_launch_us": round(1e3 * ms / n, 2), "flops_per_launch": work / n,
                           "event_stride": args.event_stride}  # fmt: skip
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_bench_b108.json")
        if os.path.exists(pmc_path) and B == 108 and (T, NCLS, args.model) == (1, 2, "prithvi_eo_v1_100"):
            # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
            # (tools: see DESIGN.md 6); FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md
            pmc = json.load(open(pmc_path))
            key = KERNEL_OF.get(dom, dom).replace(",", ", ")
            ent = next((v for k, v in pmc.items() if k.replace(" ", "") == key.replace(" ", "")), None)
            if ent:
                out["roofline"]["traffic"] = round((2 * ent["fetch_kb_raw"] + ent["write_kb"]) * 1024)
                out["roofline"]["traffic_note"] = "bytes/launch, offline PMC passes (profiles/r01_pmc_bench_b108.json)"
        out["roofline_timed_region"] = timed
        out["roofline_all"] = allk  # separate untimed pass (3 steps) with events on every MFMA entry point
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
