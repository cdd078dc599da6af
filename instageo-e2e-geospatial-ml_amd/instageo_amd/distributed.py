"""Data parallelism for the Prithvi path: one process per GPU, RCCL over xGMI through ``torch.distributed``.

The reference has no distributed code: multi-GPU only happens implicitly through Lightning's DDP
(``instageo/model/pipeline_utils.py:368-374``).  Here the exchange is explicit:

* training: chips are sharded over ranks; the only exchange per step is the gradient mean.  Because all
  gradients live in ONE flat buffer laid out in forward order, "buckets" are contiguous slices.  The engine
  reports slices as their gradients become final during backward (head first, then blocks L-1..0, patch
  embedding last) and :class:`GradSync` launches one asynchronous all-reduce per >= ``bucket_bytes`` slice so
  that communication overlaps the remaining backward kernels.  xGMI is point-to-point (7 links/GPU): a few
  large buckets (default 32 MiB ~ one ViT block) keep every link busy without per-call latency dominating.
* metrics: k x k int64 confusion matrices and (loss_sum, count) pairs are summed across ranks at epoch end
  (a deliberate fix of the reference's rank-local metrics).
* inference: windows/chips are partitioned contiguously by rank with no data-path collective; a final gather
  of the per-rank int8 class maps (or only of the counters) goes to rank 0.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


# CUs asked for RCCL while the data-parallel backward runs (IG_RESERVED_CUS overrides).  A SOFT reservation for the tile-walking
# GEMMs: ig_tile_grid launches the fewest workgroups that keep the number of rounds and gives CUs up only when that is free
# (at the benchmark's 84 x 3 / x 9 / x 12 tile counts one CU less is one round more: +25...100 % per launch); the split-K weight
# gradients leave 4-40 CUs idle by construction.  IG_RESERVED_STRICT=1 makes it strict.
DEFAULT_RESERVED_CUS = 8


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torchrun contract).

    Returns (rank, local_rank, world_size).  With WORLD_SIZE unset or 1 nothing is initialised.
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # functional testing of the multi-rank path on a one-GPU box: IG_DIST_BACKEND=gloo with LOCAL_RANK=0 on every rank
    backend = backend or os.environ.get("IG_DIST_BACKEND") or None
    # IG_DIST_FORCE=1: initialise the process group (and run every collective of the data-parallel path) with ONE rank too -- the
    # pre-flight of the RCCL code path on a one-GPU box, where every collective degenerates to a copy (tests/test_gpu_data_parallel.py,
    # `IG_DIST_FORCE=1 python bench.py`)
    if (world > 1 or os.environ.get("IG_DIST_FORCE") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # this image's host driver only supports dmabuf IPC: without it RCCL's (and torch's) cross-process device-memory
        # handles fail with "hipIpcGetMemHandle: invalid argument" (it is exported by the launch environment; set here too
        # so that a bare `python -m torch.distributed.run` works)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def dp_active() -> bool:
    """True when the collectives of the data-parallel path have to run: more than one rank, or a process group of ONE rank that
    was initialised with ``IG_DIST_FORCE=1`` (see :func:`init_from_env`)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("IG_DIST_FORCE") == "1"


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous partition of ``n_items`` units: the first ``n % world`` ranks get one extra unit."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradSync:
    """Bucketed, overlapped gradient averaging over a flat gradient buffer.

    ``ready(lo, hi)`` is called (by ``SegEngine.on_grad_ready``) with flat index ranges whose gradients are
    final, in descending address order; adjacent ranges are merged and flushed as one asynchronous
    all-reduce once they reach ``bucket_bytes``.  ``wait()`` flushes the remainder and makes the current
    stream wait for all reductions (then divides by the world size when the backend has no AVG).
    """

    def __init__(self, get_grad: Callable[[], torch.Tensor], bucket_bytes: int = 32 << 20, group=None, scale_in_optimizer: bool = False):
        self.get_grad = get_grad
        self.bucket_bytes = bucket_bytes
        self.group = group
        self.handles: List = []
        self.ranges: List[Tuple[int, int]] = []
        self.cur: Optional[Tuple[int, int]] = None
        self.launched: List[Tuple[int, int]] = []  # for tests / introspection
        # SUM everywhere (ReduceOp.AVG is not available on every backend / RCCL build); the 1/world factor is applied
        # either here after the wait, or for free inside the AdamW kernel (scale_in_optimizer=True)
        self.scale_in_optimizer = scale_in_optimizer
        self.time_buckets = False  # bench.py diagnostics: device time of every bucket's all-reduce (adds a sync per bucket)
        self._bucket_ms: List[Tuple[Tuple[int, int], float]] = []

    def ready(self, lo: int, hi: int) -> None:
        if not dp_active() or hi <= lo:
            return
        if self.cur is not None and hi == self.cur[0]:
            self.cur = (lo, self.cur[1])
        elif self.cur is not None and lo == self.cur[1]:
            self.cur = (self.cur[0], hi)
        else:
            self._flush()
            self.cur = (lo, hi)
        if (self.cur[1] - self.cur[0]) * 4 >= self.bucket_bytes:
            self._flush()

    def _flush(self) -> None:
        if self.cur is None:
            return
        lo, hi = self.cur
        self.cur = None
        g = self.get_grad()[lo:hi]
        if self.time_buckets and g.is_cuda:  # diagnostic mode: synchronous, timed (NOT overlapped)
            torch.cuda.synchronize()
            import time as _time

            t0 = _time.perf_counter()
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            torch.cuda.synchronize()
            self._bucket_ms.append(((lo, hi), 1e3 * (_time.perf_counter() - t0)))
        else:
            self.handles.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.ranges.append((lo, hi))
        self.launched.append((lo, hi))

    def bucket_times(self) -> List[Tuple[Tuple[int, int], float]]:
        """[(flat range, milliseconds)] of the buckets reduced while ``time_buckets`` was set; clears the record."""
        out, self._bucket_ms = self._bucket_ms, []
        return out

    def wait(self) -> None:
        if not dp_active():
            return
        self._flush()
        for h in self.handles:
            h.wait()
        if not self.scale_in_optimizer:
            g = self.get_grad()
            w = float(world_size())
            for lo, hi in self.ranges:
                g[lo:hi].div_(w)
        self.handles.clear()
        self.ranges.clear()


class ShardedGradSync:
    """Gradient exchange as bucketed REDUCE-SCATTER + parameter ALL-GATHER around a sharded optimizer step.

    The flat optimizer range ``[lo, hi)`` is cut into the same contiguous buckets as :class:`GradSync` (ranges arrive
    head-first in descending address order while backward runs), but a bucket's length is kept a multiple of
    ``world * ALIGN`` elements and each bucket is reduce-scattered: rank r receives the summed gradients of slice r of every
    bucket into its ``owned`` buffer.  ``step(adam)`` then runs AdamW on the owned slices only -- 1/world of the optimizer
    work and of the moment buffers per rank -- and all-gathers every bucket's updated fp32 parameters, bucket by bucket, so
    the gathers of the first buckets overlap the AdamW launches of the later ones.  Bytes on the wire equal one all-reduce
    (reduce-scatter + all-gather IS the ring all-reduce, split around the optimizer).  The tail of the range that does not
    fill a multiple of ``world * ALIGN`` (< that many elements) is all-reduced and updated redundantly by every rank.

    ``adam(param, grad, m, v, index0)`` is the optimizer kernel on one contiguous slice (``FusedAdamW`` passes
    ``ig_adamw_step``; the CPU tests pass a torch restatement): ``param`` fp32 slice of the flat buffer (updated in place),
    ``grad`` the SUMMED gradient slice (the 1/world factor is the optimizer's grad scale), ``m`` / ``v`` moment slices owned by
    this object, ``index0`` the slice's offset in the flat buffer.
    """

    ALIGN = 64

    def __init__(self, get_grad: Callable[[], torch.Tensor], get_flat: Callable[[], torch.Tensor], lo: int, hi: int,
                 bucket_bytes: int = 32 << 20, group=None):
        self.get_grad, self.get_flat, self.lo, self.hi = get_grad, get_flat, lo, hi
        self.bucket_bytes, self.group = bucket_bytes, group
        self.world = world_size()
        self.rank = dist.get_rank() if dp_active() else 0
        self.q = self.world * self.ALIGN
        self.cur: Optional[Tuple[int, int]] = None
        self.plan: List[Tuple[int, int, int]] = []      # (lo, hi, owned elements before it), fixed after the first step
        self._step_buckets: List[Tuple[int, int]] = []  # buckets flushed in the current step
        self._handles: List = []
        self._own: List[torch.Tensor] = []              # per bucket: this rank's slice of the summed gradient
        self._m: List[torch.Tensor] = []                # per bucket: AdamW moments of the owned slice
        self._v: List[torch.Tensor] = []
        self.tail: Optional[Tuple[int, int]] = None     # replicated remainder (all-reduced)
        self.m_tail: Optional[torch.Tensor] = None
        self.v_tail: Optional[torch.Tensor] = None
        self._tail_handle = None
        self._frozen = False
        self.launched: List[Tuple[int, int]] = []
        self.master_complete = True      # False after a step that published only the bf16 operand copy (see step / gather_master)
        self._small = None               # index tables of the fp32-read parameters (see _sync_small)
        self.time_exposed = False        # bench.py diagnostics: events around every wait of the step
        self._exposed: List = []
        self._pending: List = []         # (bucket lo, handle) of all-gathers whose wait was deferred to the next reader (wait_params)

    # ---- during backward -------------------------------------------------------------------------------------------
    def ready(self, lo: int, hi: int) -> None:
        lo, hi = max(lo, self.lo), min(hi, self.hi)
        if hi <= lo:
            return
        if self.cur is not None and hi == self.cur[0]:
            self.cur = (lo, self.cur[1])
        elif self.cur is None:
            self.cur = (lo, hi)
        else:
            raise RuntimeError("ShardedGradSync: gradient ranges must arrive adjacent, in descending address order")
        if (self.cur[1] - self.cur[0]) * 4 >= self.bucket_bytes:
            self._flush(final=False)

    def _flush(self, final: bool) -> None:
        if self.cur is None:
            return
        lo, hi = self.cur
        span = ((hi - lo) // self.q) * self.q
        if span > 0:
            blo = hi - span
            g = self.get_grad()
            n = span // self.world
            i = len(self._step_buckets)
            if self._frozen:
                if i >= len(self.plan) or self.plan[i][:2] != (blo, hi):
                    raise RuntimeError("ShardedGradSync: the bucket sequence changed between steps (moment slices would be misassigned)")
            else:  # first step: one (gradient slice, m, v) triple per bucket, kept for the life of the object
                self.plan.append((blo, hi, sum(t.numel() for t in self._own)))
                self._own.append(torch.zeros(n, dtype=torch.float32, device=g.device))
                self._m.append(torch.zeros(n, dtype=torch.float32, device=g.device))
                self._v.append(torch.zeros(n, dtype=torch.float32, device=g.device))
            self._handles.append(self._reduce_scatter(self._own[i], g[blo:hi], n))
            self._step_buckets.append((blo, hi))
            self.launched.append((blo, hi))
            hi = blo
        self.cur = (lo, hi) if hi > lo else None
        if final and self.cur is not None:
            tlo, thi = self.cur
            self.cur = None
            if self.tail is None:
                g = self.get_grad()
                self.tail = (tlo, thi)
                self.m_tail = torch.zeros(thi - tlo, dtype=torch.float32, device=g.device)
                self.v_tail = torch.zeros_like(self.m_tail)
            elif self.tail != (tlo, thi):
                raise RuntimeError("ShardedGradSync: the replicated tail changed between steps")
            self._tail_handle = dist.all_reduce(self.get_grad()[tlo:thi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    # Backends without a native reduce-scatter / all-gather for the tensor's device (gloo with device tensors on the one-GPU
    # test box) take an equivalent form: all-reduce of the bucket + copy of the own slice, and one broadcast per slice.  The
    # probe result is the same on every rank (same backend, same device kind), so the collective sequences stay matched.
    _native = {"rs": None, "ag": None}

    class _Done:
        def wait(self):
            return None

    def _reduce_scatter(self, out: torch.Tensor, bucket: torch.Tensor, n: int):
        if self._native["rs"] is not False:
            try:
                h = dist.reduce_scatter_tensor(out, bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._native["rs"] = True
                return h
            except (RuntimeError, NotImplementedError):
                if self._native["rs"]:
                    raise
                self._native["rs"] = False
        h = dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        class _Copy:
            def wait(_self):
                h.wait()
                out.copy_(bucket[self.rank * n : (self.rank + 1) * n])

        return _Copy()

    def _all_gather(self, full: torch.Tensor, send: torch.Tensor):
        if self._native["ag"] is not False:
            try:
                h = dist.all_gather_into_tensor(full, send, group=self.group, async_op=True)
                self._native["ag"] = True
                return h
            except (RuntimeError, NotImplementedError):
                if self._native["ag"]:
                    raise
                self._native["ag"] = False
        n = send.numel()
        hs = [dist.broadcast(full[r * n : (r + 1) * n], src=r, group=self.group, async_op=True) for r in range(self.world)]

        class _All:
            def wait(_self):
                for x in hs:
                    x.wait()

        return _All()

    # ---- after backward --------------------------------------------------------------------------------------------
    def _timed_wait(self, handle, kind: str) -> None:
        """``handle.wait()``; in the instrumented step also the time the compute stream stands still in it (two events around
        the wait on the current stream: nothing else is enqueued between them, so their distance IS the exposed communication)."""
        if self.time_exposed and torch.cuda.is_available():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            handle.wait()
            b.record()
            self._exposed.append((kind, a, b))
        else:
            handle.wait()

    def exposed_ms(self) -> dict:
        """{"rs": ms, "ag": ms, "tail": ms}: exposed (un-overlapped) communication of the steps run with ``time_exposed`` set;
        clears the record.  Synchronises the device."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        out = {"rs": 0.0, "ag": 0.0, "tail": 0.0}
        for kind, a, b in self._exposed:
            out[kind] += a.elapsed_time(b)
        self._exposed = []
        return out

    def step(self, adam: Callable[[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, int], None],
             gather: Optional[Sequence[torch.Tensor]] = None, small_ranges: Optional[Sequence[Tuple[int, int]]] = None,
             defer: bool = False) -> None:
        """Finish the exchange, run ``adam`` on the owned slices (and the replicated tail), publish the result.

        ``gather`` None: the updated fp32 parameters are all-gathered (4 bytes per parameter).  ``gather`` = flat tensors indexed
        like the parameter buffer (the bf16 operand copy: hi, and lo in the split-precision mode) -- what the other ranks compute
        with -- : THOSE are all-gathered instead (2 bytes per parameter; ``adam`` has written this rank's slice of them), the fp32
        masters stay sharded (:meth:`gather_master` completes them, e.g. before a checkpoint), and the parameters the kernels read
        in fp32 (``small_ranges``: biases, norm affine, classifier -- a few hundred KB) are exchanged by ONE small all-reduce in which
        every rank contributes the elements it owns.  ``defer`` (with ``gather``): return without waiting for the all-gathers;
        :meth:`wait_params` must then run in front of every reader of the gathered tensors."""
        self._flush(final=True)
        flat = self.get_flat()
        # every step: the bucket sequence must be the planned one BEFORE anything is updated (a missing ready() range would leave
        # some parameters without their update while the others step)
        if self._frozen and len(self._step_buckets) != len(self.plan):
            raise RuntimeError(f"ShardedGradSync: {len(self._step_buckets)} of {len(self.plan)} planned buckets were reported this step")
        if not self._frozen:
            covered = sum(b - a for a, b, _ in self.plan) + (0 if self.tail is None else self.tail[1] - self.tail[0])
            if covered != self.hi - self.lo:
                raise RuntimeError(f"ShardedGradSync: buckets cover {covered} of {self.hi - self.lo} gradient elements "
                                   "(every trainable range must be reported through ready())")
        self.wait_params()  # (a step without a forward in between: nothing of the previous publication may still be in flight)
        for i, (blo, bhi) in enumerate(self._step_buckets):
            self._timed_wait(self._handles[i], "rs")
            n = (bhi - blo) // self.world
            s0 = blo + self.rank * n
            adam(flat[s0 : s0 + n], self._own[i], self._m[i], self._v[i], s0)
        if self._tail_handle is not None:
            self._timed_wait(self._tail_handle, "tail")
            tlo, thi = self.tail
            adam(flat[tlo:thi], self.get_grad()[tlo:thi], self.m_tail, self.v_tail, tlo)
            self._tail_handle = None
        if gather is not None and small_ranges:  # before the all-gathers: a collective queued behind them would wait for all of them
            self._sync_small(flat, small_ranges)
        # The all-gathers go out in ASCENDING address order -- the order in which the next forward pass reads the parameters
        # (patch embed, block 0, 1, ...): with ``defer`` the wait for a bucket is left to the first kernel that reads it
        # (wait_params, called by the engine in front of every Block), so the exchange rides under the next step's normalise, patch
        # embed and earlier blocks instead of standing between two steps.
        gathers = []
        for blo, bhi in sorted(self._step_buckets):
            n = (bhi - blo) // self.world
            s0 = blo + self.rank * n
            for t in ([flat] if gather is None else gather):
                send = t[s0 : s0 + n].clone()  # (not the in-place form: output and input of the gather do not alias)
                gathers.append((blo, self._all_gather(t[blo:bhi], send)))
        if defer and gather is not None:
            self._pending = gathers
        else:
            for _, h in gathers:
                self._timed_wait(h, "ag")
        self._frozen = True
        self.master_complete = gather is None
        self._handles.clear()
        self._step_buckets.clear()

    def wait_params(self, upto: Optional[int] = None) -> None:
        """Wait (on the current stream) for the deferred all-gathers of every bucket that starts below flat offset ``upto`` (None:
        all of them).  The engine calls it with the end offset of the parameters it is about to read (``SegEngine.param_wait``)."""
        while self._pending and (upto is None or self._pending[0][0] < upto):
            _, h = self._pending.pop(0)
            self._timed_wait(h, "ag")

    def _sync_small(self, flat: torch.Tensor, small_ranges: Sequence[Tuple[int, int]]) -> None:
        """fp32 parameters inside the sharded buckets that the kernels read directly: every rank fills a compact buffer with the
        elements it owns (zeros elsewhere), one all-reduce (SUM) completes it, the result goes back into the flat buffer."""
        if self._small is None:
            idx_all, pos_own, idx_own, at = [], [], [], 0
            owned = [(blo + self.rank * ((bhi - blo) // self.world), blo + (self.rank + 1) * ((bhi - blo) // self.world)) for blo, bhi, _ in self.plan]
            for lo, hi in small_ranges:
                for blo, bhi, _ in self.plan:  # the part of the range inside sharded buckets (the replicated tail needs nothing)
                    a, b = max(lo, blo), min(hi, bhi)
                    if b <= a:
                        continue
                    idx_all.append(torch.arange(a, b))
                    for olo, ohi in owned:
                        c, d = max(a, olo), min(b, ohi)
                        if d > c:
                            idx_own.append(torch.arange(c, d))
                            pos_own.append(torch.arange(at + c - a, at + d - a))
                    at += b - a
            dev = flat.device
            cat = lambda xs: (torch.cat(xs) if xs else torch.zeros(0, dtype=torch.int64)).to(dev)  # noqa: E731
            self._small = (cat(idx_all), cat(idx_own), cat(pos_own), torch.zeros(at, dtype=torch.float32, device=dev))
        idx_all, idx_own, pos_own, buf = self._small
        if buf.numel() == 0:
            return
        buf.zero_()
        buf[pos_own] = flat[idx_own]
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        flat[idx_all] = buf

    def gather_master(self) -> None:
        """Complete the fp32 master parameters on every rank (after steps that published only the bf16 operand copy): one fp32
        all-gather per bucket.  Collective -- every rank calls it (e.g. at epoch end, before rank 0 writes a checkpoint)."""
        self.wait_params()
        if not dp_active() or self.master_complete or not self.plan:
            return
        flat = self.get_flat()
        hs = []
        for blo, bhi, _ in self.plan:
            n = (bhi - blo) // self.world
            s0 = blo + self.rank * n
            hs.append(self._all_gather(flat[blo:bhi], flat[s0 : s0 + n].clone()))
        for h in hs:
            h.wait()
        self.master_complete = True

    def optimizer_elements(self) -> int:
        """fp32 elements of moment state held by this rank (about 1/world of the replicated optimizer's)."""
        return sum(t.numel() for t in self._m) + (0 if self.m_tail is None else self.m_tail.numel())

    # ---- checkpoint / resume of the sharded moments ----------------------------------------------------------------
    def state_dict(self) -> dict:
        """This rank's optimizer state: the bucket plan and the AdamW moments of the owned slices (+ the replicated tail)."""
        return {"world": self.world, "rank": self.rank, "plan": list(self.plan), "tail": self.tail,
                "m": [t.detach().cpu().clone() for t in self._m], "v": [t.detach().cpu().clone() for t in self._v],
                "m_tail": None if self.m_tail is None else self.m_tail.detach().cpu().clone(),
                "v_tail": None if self.v_tail is None else self.v_tail.detach().cpu().clone()}

    def load_state_dict(self, sd: dict) -> None:
        """Restore :meth:`state_dict` of the SAME rank / world size / bucket plan (a plan exists after the first step; before it,
        the saved plan is adopted and checked against the buckets as they arrive)."""
        if sd["world"] != self.world or sd["rank"] != self.rank:
            raise RuntimeError(f"ShardedGradSync: state of rank {sd['rank']}/{sd['world']} loaded into rank {self.rank}/{self.world}")
        if self._frozen and [tuple(p) for p in sd["plan"]] != [tuple(p) for p in self.plan]:
            raise RuntimeError("ShardedGradSync: the saved bucket plan differs from the current one")
        dev = self.get_flat().device
        self.plan = [tuple(p) for p in sd["plan"]]
        self.tail = None if sd["tail"] is None else tuple(sd["tail"])
        self._m = [t.to(dev).clone() for t in sd["m"]]
        self._v = [t.to(dev).clone() for t in sd["v"]]
        self._own = [torch.zeros_like(t) for t in self._m]
        self.m_tail = None if sd["m_tail"] is None else sd["m_tail"].to(dev).clone()
        self.v_tail = None if sd["v_tail"] is None else sd["v_tail"].to(dev).clone()
        self._frozen = True

    def full_moments(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(m, v) over the whole optimizer range [lo, hi) on EVERY rank (collective: one all-gather per bucket and moment) -- the
        layout of the replicated optimizer, for a world-size independent checkpoint."""
        dev = self.get_flat().device
        m = torch.zeros(self.hi - self.lo, dtype=torch.float32, device=dev)
        v = torch.zeros_like(m)
        for i, (blo, bhi, _) in enumerate(self.plan):
            for full, part in ((m, self._m[i]), (v, self._v[i])):
                if dp_active():
                    self._all_gather(full[blo - self.lo : bhi - self.lo], part.clone()).wait()
                else:
                    full[blo - self.lo : bhi - self.lo] = part
        if self.tail is not None:
            m[self.tail[0] - self.lo : self.tail[1] - self.lo] = self.m_tail
            v[self.tail[0] - self.lo : self.tail[1] - self.lo] = self.v_tail
        return m, v


def attach_data_parallel(module, bucket_bytes: int = 32 << 20):
    """Wire a :class:`GradSync` into a ``PrithviSegmentationModule`` (fused path) and broadcast rank 0's
    parameters/buffers so all replicas start equal (Lightning/DDP semantics)."""
    if not dp_active():
        return None
    net = module.net
    dist.broadcast(net.store.flat, src=0)
    for t in net._buffers_flat.values():
        dist.broadcast(t, src=0)
    net.params_changed()
    # The GEMM kernels are persistent, one workgroup per CU: with every CU pinned, RCCL's all-reduce kernels (launched from the
    # backward as buckets become final) would wait for a whole GEMM to drain.  Leave a few CUs free while world > 1.
    from . import ops

    if net.store.flat.is_cuda and "IG_RESERVED_CUS" not in os.environ and ops.reserved_cus() == 0:
        ops.set_reserved_cus(DEFAULT_RESERVED_CUS)
    opt = module.optimizer()
    opt.set_grad_scale(1.0 / world_size())
    # IG_DP_MODE: "zero1" (default) = reduce-scatter + AdamW on the owned 1/world slices + all-gather of the parameters;
    # "allreduce" = one all-reduce per bucket and the replicated optimizer (round 1-2 behaviour)
    mode = os.environ.get("IG_DP_MODE", "zero1")
    if mode == "zero1" and hasattr(opt, "attach_sharded"):
        sync = ShardedGradSync(lambda: net.store.ensure_grad(), lambda: net.store.flat, opt.lo, opt.hi, bucket_bytes)
        opt.attach_sharded(sync)
        net.engine.on_grad_ready = sync.ready
        net.engine.master_sync = sync.gather_master
        net.engine.master_complete = lambda: sync.master_complete  # state_dict() refuses to read sharded masters (see PrithviSeg.state_dict)
        net.engine.param_wait = sync.wait_params                   # deferred all-gather waits, in front of each Block of the next forward
        net.store.pending_from = lambda: (sync._pending[0][0] if sync._pending else None)  # ParamStore.w() refuses to read under an in-flight gather
        module.grad_sync = None  # the exchange is finished inside the optimizer step
        return sync
    sync = GradSync(lambda: net.store.ensure_grad(), bucket_bytes, scale_in_optimizer=True)
    net.engine.on_grad_ready = sync.ready
    module.grad_sync = sync.wait
    return sync


def reduce_confusion(matrix: torch.Tensor) -> torch.Tensor:
    """Sum the k x k int64 confusion matrix over ranks (C2 of SURVEY.md 2.2)."""
    if dp_active():
        dist.all_reduce(matrix, op=dist.ReduceOp.SUM)
    return matrix


def reduce_loss_stats(stats: torch.Tensor) -> torch.Tensor:
    """Sum (loss_sum, count) over ranks."""
    if dp_active():
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return stats


def gather_class_maps(local: torch.Tensor, counts: Sequence[int], dst: int = 0) -> Optional[torch.Tensor]:
    """Gather per-rank int8 class maps (n_r, H, W) to ``dst`` (C3 of SURVEY.md 2.2).  ``counts[r]`` = n_r."""
    world = world_size()
    if not dp_active():
        return local
    rank = dist.get_rank()
    nmax = max(counts)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, outs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[: counts[r]] for r, o in enumerate(outs)], dim=0)
