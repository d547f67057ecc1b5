"""Minimal one-process-per-GPU trainer that stands where PyTorch-Lightning's Trainer stands in the reference
(SURVEY.md L4): it owns the optimiser/scheduler, drives `training_step`, and implements the data-parallel part.

Data parallelism: image pairs are independent, so ranks shard the batch; align / refine / teacher are
gradient-free replicas with ZERO communication.  The student's gradients of the three backward passes of a step are
accumulated into one flat fp32 buffer (every `p.grad` is a view into it) and all-reduced ONCE per step with RCCL,
averaged over ranks -- the reference's DDP does the same reduction three times per step (once per manual_backward).
On xGMI (7 point-to-point links per GPU) one 342 MB all-reduce is bandwidth-bound; the flat buffer is laid out in
gradient-readiness order (decode heads + MiT stage 4, stage 3, stage 2, stage 1) and, during the LAST backward pass of
the step, each finished range is put on the RCCL stream in `bucket_mb` pieces while the backward of the earlier stages
is still running (marks in seg.MixVisionTransformer.forward_features); the remainder follows when the pass ends.
BatchNorm layers become SyncBatchNorm when the config says `sync_batchnorm: True` (student AND teacher, D9).
"""
import os
from typing import List

import torch
import torch.distributed as dist
import torch.nn as nn

from .params import mark_grad_sink


class LinearWarmupPolynomialLR(torch.optim.lr_scheduler.LRScheduler):
    """helpers/lr_scheduler.py:10-57: linear warm-up from warmup_ratio*lr over warmup_iters, then polynomial decay to
    min_lr at max_steps."""

    def __init__(self, optimizer, max_steps: int = None, warmup_iters: int = 1500, warmup_ratio: float = 1e-6,
                 power=0.9, min_lr=0., last_epoch=-1):
        self.max_updates, self.warmup_iters, self.warmup_ratio = max_steps, warmup_iters, warmup_ratio
        self.power, self.min_lr = power, min_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self) -> List[float]:
        t = self.last_epoch
        if t < self.warmup_iters:
            k = (1 - t / self.warmup_iters) * (1 - self.warmup_ratio)
            return [lr * (1 - k) for lr in self.base_lrs]
        coeff = (1 - (t - self.warmup_iters) / float(self.max_updates - self.warmup_iters)) ** self.power
        return [(lr - self.min_lr) * coeff + self.min_lr for lr in self.base_lrs]


class ValEveryNSteps:
    """helpers/callbacks.py: validation every N steps (evaluation is out of scope here; kept so configs parse)."""

    def __init__(self, every_n_steps):
        self.every_n_steps = every_n_steps


class FlatGradBuffer:
    """All trainable parameters' gradients as views into one contiguous fp32 tensor.

    Limitation (vs. the reference, where a parameter that receives no gradient keeps `.grad = None` and is skipped by the
    optimizer): every view is persistent and zero-filled, so a trainable parameter that is never touched in a step still
    takes its AdamW weight-decay / moment update with a zero gradient.  No module of the hot path has such a parameter
    (every trainable tensor of MiT / DAFormer / SegFormer heads is on all three backward passes); a custom head with an
    unused branch should freeze it (`requires_grad_(False)`).

    `groups`: [(tag, [params])] in the order the gradients become FINAL during a backward pass
    (uda.grad_ready_groups): the buffer is laid out in that order, so `on_ready(tag)` -- called from the backward pass
    itself through seg._GradMark -- can put the range of a finished group on the wire (async all-reduce in `bucket_mb`
    pieces on the RCCL stream) while the backward of the earlier MiT stages is still running.  `all_reduce_mean()`
    reduces whatever has not been released, waits for everything and divides by the world size."""

    ALIGN = 64          # elements: every view starts on a 256-byte boundary (16-byte vector stores in csrc/reduce.hip)

    def __init__(self, params, groups=None, bucket_mb=64):
        self.params = [p for p in params if p.requires_grad]
        order, self.ranges = [], {}
        known = {id(p) for p in self.params}
        pad = lambda k: (k + self.ALIGN - 1) // self.ALIGN * self.ALIGN  # noqa: E731
        off, placed = 0, set()
        for tag, ps in (groups or []):
            start = off
            for p in ps:
                if id(p) in known and id(p) not in placed:
                    order.append(p)
                    placed.add(id(p))
                    off += pad(p.numel())
            if off > start:
                self.ranges[tag] = (start, off)
        rest = [p for p in self.params if id(p) not in placed]
        order += rest
        n = off + sum(pad(p.numel()) for p in rest)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in order:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            mark_grad_sink(p)                       # backward kernels may accumulate into this view directly
            off += pad(p.numel())
        self.bucket_mb = bucket_mb
        self._works, self._released = [], []
        self._order, self.flat2 = order, None
        self._comm = self._comm2 = self._stream = None
        self._active = 0                                # which buffer the parameters' .grad views point into right now
        self._in_capture, self.captured_ranges, self._direct_pending = [], (), False
        self.overlapped_elements = 0

    def use_direct(self, comm, stream=None, comm2=None):
        """Reduce over `comm` (refign_amd/rccl.py DirectComm: ncclAllReduce on the stream WE choose) on `stream`, a stream of
        the reduce's own that waits for the producing stream at every release -- instead of torch's process group (whose
        collectives hop to the group's stream and back: 15 ms per step for the six 64 MB buckets of a 1-rank group, measured,
        profiles/r04_ddp_rehearsal.txt).  Event record / wait / ncclAllReduce are all capturable, so the SAME release works
        inside the hipGraph capture of the last backward pass: the all-reduce of the decode head / stage 4 / stage 3 ranges
        becomes a branch of the graph next to the backward of the earlier stages (`end_capture`, `replayed`).
        `comm2`: a second communicator for the SECOND buffer (the mixed pass running next to the source pass accumulates
        there): the sum over ranks is linear, so the two buffers are reduced separately -- the first one whole, as soon as the
        source pass is over and next to the mixed pass (`reduce_first_now`), the second one range by range from inside the
        mixed pass's backward -- and added afterwards.  Twice the bytes on the links, almost all of them hidden; two
        communicators because the two reduces run on unordered streams and one communicator's collectives must be issued in
        the same order on every rank."""
        self._comm, self._stream, self._comm2 = comm, stream, comm2

    def zero(self):
        self.flat.zero_()
        if self.flat2 is not None:
            self.flat2.zero_()
        self._in_capture = []

    def into_second(self):
        """Context manager: inside it every parameter's `.grad` is its view of a SECOND flat buffer.  Used around the
        hipGraph capture of a backward pass that is to run concurrently with another one (the captured kernels keep the
        addresses they saw): two passes accumulating into one buffer from two streams would race on every
        read-modify-write; `merge_second()` adds the second buffer to the first before the optimiser looks."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            if self.flat2 is None:
                self.flat2 = torch.zeros_like(self.flat)
            saved, off = [], 0
            pad = lambda k: (k + self.ALIGN - 1) // self.ALIGN * self.ALIGN  # noqa: E731
            for p in self._order:
                saved.append(p.grad)
                p.grad = self.flat2[off:off + p.numel()].view_as(p)
                off += pad(p.numel())
            self._active = 1
            try:
                yield
            finally:
                self._active = 0
                for p, g_ in zip(self._order, saved):
                    p.grad = g_
        return ctx()

    def merge_second(self):
        if self.flat2 is not None:
            self.flat.add_(self.flat2)

    def _buffer(self, k):
        return self.flat if k == 0 else self.flat2

    def _reduce_range(self, a, b, k=0):
        step = max(1, int(self.bucket_mb * 1024 * 1024 // 4))
        buf = self._buffer(k)
        comm = self._comm if k == 0 else self._comm2
        if comm is not None:
            import contextlib
            on = contextlib.nullcontext()
            if self._stream is not None:
                self._stream.wait_stream(torch.cuda.current_stream(buf.device))           # the gradients queued so far
                on = torch.cuda.stream(self._stream)
            with on:
                for i in range(a, b, step):
                    comm.all_reduce_(buf[i:min(i + step, b)])
            self._direct_pending = True
            return
        for i in range(a, b, step):
            self._works.append(dist.all_reduce(buf[i:min(i + step, b)], op=dist.ReduceOp.SUM, async_op=True))

    def on_ready(self, tag, capturing=None):
        """A group's gradients are final (called from inside the last backward pass into the buffer the views point to)."""
        r = self.ranges.get(tag)
        if r is None or not (dist.is_available() and dist.is_initialized()):
            return
        r = (self._active, *r)
        if r in self._released or r in self._in_capture:
            return
        if capturing is None:
            capturing = self.flat.is_cuda and torch.cuda.is_current_stream_capturing()
        if (self._comm if self._active == 0 else self._comm2) is None and (capturing or self._active == 1):
            # (a collective of torch's process group is not put into a capture here, and the second buffer is only reduced
            # on its own when it has a communicator of its own: the tail takes care of both)
            return
        # (torch's NCCL process group orders an async collective after the work queued so far on the CURRENT stream --
        # inside the autograd engine that is the stream of the backward kernels that produced these gradients)
        (self._in_capture if capturing else self._released).append(r)
        self._reduce_range(r[1], r[2], r[0])

    def reduce_first_now(self):
        """The FIRST buffer is final (the source pass is over; the rest of the step accumulates into the second one): put it on
        the wire whole, behind what is queued on the current stream.  Needs both communicators (see use_direct)."""
        if self._comm is None or self._comm2 is None or not (dist.is_available() and dist.is_initialized()):
            return False
        r = (0, 0, self.flat.numel())
        if r not in self._released:
            self._released.append(r)
            self._reduce_range(0, self.flat.numel(), 0)
        return True

    def end_capture(self):
        """INSIDE the capture, after the backward pass: join the reduce stream; `captured_ranges` <- the ranges the graph
        reduces on every replay (graphs.GraphedStep hands them back to `replayed` after each replay)."""
        if self._in_capture and self._stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._stream)
        self.captured_ranges, self._in_capture = tuple(self._in_capture), []
        return self.captured_ranges

    def replayed(self, ranges):
        """A graph that contains the all-reduce of `ranges` has just been replayed in this step."""
        for r in ranges or ():
            if r not in self._released:
                self._released.append(r)

    def _reduce_gaps(self, k):
        pos, n = 0, self.flat.numel()
        for _, a, b in sorted(r for r in self._released if r[0] == k):
            if a > pos:
                self._reduce_range(pos, a, k)
            pos = max(pos, b)
        if pos < n:
            self._reduce_range(pos, n, k)

    def all_reduce_mean(self, bucket_mb=None):
        """Everything the passes have not released, then: wait, add the second buffer, divide by the world size."""
        from .bn import data_parallel
        if not (dist.is_available() and dist.is_initialized()) or not data_parallel():
            # no group, or a group of ONE rank (torchrun --nproc-per-node 1): the sum over one rank is the identity and the
            # mean divides by 1 -- no collective is issued (round 4 sent the 343 MB buffer through six bucketed all-reduces of
            # the 1-rank group anyway: 144.9 -> 159.9 ms/step, the all-reduce kernels of torch's process group running on its
            # own stream next to the three busy compute streams).  RFN_DDP_REHEARSAL=1 makes a 1-rank group exchange for real.
            self.merge_second()
            self._released = []
            return
        if bucket_mb is not None:
            self.bucket_mb = bucket_mb
        world = dist.get_world_size()
        # the second buffer travels on its own as soon as ANY range has been released before this point: adding it into the first
        # buffer here would race with that range's all-reduce in flight -- and would add un-reduced values to reduced ones
        # (the whole-first-buffer and second-buffer releases were the cases round 4 handled)
        separate = self.flat2 is not None and len(self._released) > 0
        if not separate:
            self.merge_second()                       # one sum on the wire: what the second buffer holds goes with the first
        self._reduce_gaps(0)
        if separate:
            self._reduce_gaps(1)
        for w in self._works:
            w.wait()
        if self._direct_pending and self._stream is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._stream)
        self._direct_pending = False
        if separate:
            self.merge_second()
        self.overlapped_elements = sum(b - a for _, a, b in self._released)     # diagnostics / tests
        self.overlapped_fraction = self.overlapped_elements / (self.flat.numel() * (2 if separate else 1))
        self._works, self._released = [], []
        self.flat.div_(world)


class StallGuard:
    """A multi-rank run that makes no progress for `limit` seconds (RFN_STALL_S, default 600; RFN_BENCH_STALL_S is still read) says
    where it stopped on stderr and exits with code 17 instead of hanging the node in a collective.  `note(what)` is the heartbeat:
    Trainer.step calls it around every step; bench.py adds its own phases.  One daemon thread per process."""

    def __init__(self, rank, world, limit=None):
        import threading
        import time
        self.rank, self.world = rank, world
        self.limit = float(os.environ.get("RFN_STALL_S", os.environ.get("RFN_BENCH_STALL_S", "600"))) if limit is None else float(limit)
        self.progress = [time.monotonic(), "start"]
        threading.Thread(target=self._watch, daemon=True, name="refign-stall-guard").start()

    def note(self, what):
        import time
        self.progress[0], self.progress[1] = time.monotonic(), what

    def _watch(self):
        import sys
        import time
        while True:
            time.sleep(min(5.0, max(0.2, self.limit / 4)))
            idle = time.monotonic() - self.progress[0]
            if idle > self.limit:
                print(f"refign_amd rank {self.rank}/{self.world}: no progress for {idle:.0f} s after '{self.progress[1]}'; giving up",
                      file=sys.stderr, flush=True)
                os._exit(17)


class Trainer:
    """fit-loop subset: `step(batch)` = one reference training_step including EMA, three backward passes, the single
    gradient all-reduce and the optimiser/scheduler step."""

    def __init__(self, model, sync_batchnorm=False, bucket_mb=64, fused_optimizer=True, gc_interval=None):
        self.model = model
        if gc_interval is None:
            gc_interval = int(os.environ.get("RFN_GC_INTERVAL", "100"))
        self.gc_interval, self._steps_done = gc_interval, 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        from .bn import data_parallel
        self.data_parallel = data_parallel()                 # world > 1, or a 1-rank rehearsal of it
        from .bn import ddp_mode
        self.ddp_mode = ddp_mode() if self.data_parallel else None
        if sync_batchnorm and dist.is_available() and dist.is_initialized():
            nn.SyncBatchNorm.convert_sync_batchnorm(model)   # student AND teacher BNs (reference: sync_batchnorm: True)
            # The EMA teacher runs on a side stream next to the student (and next to hipGraph replays that contain the
            # student's statistics exchanges): its BatchNorm collectives get a communicator of their own, so that the
            # two streams never interleave collectives of ONE communicator in a rank-dependent order.
            teacher = [m for n, mod in model.named_children() if n.startswith("m_") for m in mod.modules()
                       if isinstance(m, nn.SyncBatchNorm)]
            if teacher and self.data_parallel:
                group = dist.new_group()
                for m in teacher:
                    m.process_group = group
        # RCCL called directly for the statistics exchanges (refign_amd/rccl.py; RFN_DDP_MODE=direct / direct3): a collective of torch's
        # process group hops to the group's own stream and back, and with three compute streams on four hardware queues
        # that stream shares a queue with a busy one -- in the 1-rank rehearsal the teacher branch takes 138 ms instead
        # of 116.  One communicator per stream that exchanges: the student's passes on the main stream, the teacher
        # (side stream), and -- with graphed student passes -- the mixed pass, which may then run next to the source
        # pass as it does on one GPU.
        from . import bn as _bn
        _bn._DIRECT["default"] = None
        if sync_batchnorm and self.data_parallel and next(model.parameters()).is_cuda and dist.get_backend() == "nccl":
            from . import rccl
            if rccl.enabled():
                dev = next(model.parameters()).device
                try:
                    _bn._DIRECT["default"] = rccl.DirectComm(dev)
                    if teacher:
                        teacher_comm = rccl.DirectComm(dev)
                        for m in teacher:
                            m._rfn_direct = teacher_comm
                    # A THIRD communicator (RFN_DDP_MODE=direct3 only): the mixed pass on its own stream next to the source pass, as
                    # on one GPU.  Every communicator's collectives are issued from one host thread in program order on a stream
                    # of its own, the same on every rank -- but three communicators' kernels on one device have never met a
                    # second rank (bn.ddp_mode), so this is opt-in.
                    if _bn.ddp_mode() == "direct3":
                        model._mixed_comm = rccl.DirectComm(dev)
                    # The gradient reduce.  Default: torch's process group, bucketed, after the passes (with the mixed pass
                    # next to the source pass there is no cheap way to start earlier: releases from inside a captured pass are
                    # cross-stream branches of its graph, and two graphs replaying side by side pay ~25 ms for those --
                    # profiles/r04_ddp_rehearsal.txt; the reduce itself is 343 MB: ~1-5 ms of link time on 8..2 GPUs).
                    # RFN_DDP_DIRECT_REDUCE=1: a communicator and a stream of its own (FlatGradBuffer.use_direct) -- the
                    # finished ranges then travel from inside the last backward pass, eagerly and inside the captured mixed
                    # pass alike (meant for RFN_DDP_MODE=direct: passes in stream order).
                    if os.environ.get("RFN_DDP_DIRECT_REDUCE", "0") == "1":
                        self._grad_comm = rccl.DirectComm(dev)
                        # (second buffer reduced on its own -- FlatGradBuffer.use_direct -- only on request: its in-graph
                        # releases are cross-stream branches of the mixed pass's graph, and two graphs replaying next to each
                        # other pay ~25 ms for those in the rehearsal: 172.7 against 147-159 ms with one reduce at the tail)
                        self._grad_comm2 = rccl.DirectComm(dev) if getattr(model, "_mixed_comm", None) is not None and \
                            os.environ.get("RFN_DDP_TWO_BUFFER", "0") == "1" else None
                except (OSError, RuntimeError, AttributeError) as e:   # no librccl / init failed: torch's process group
                    import warnings
                    warnings.warn(f"refign_amd.trainer: direct RCCL communicators unavailable ({type(e).__name__}: {e}); "
                                  f"statistics exchanges go through torch.distributed, student passes run eagerly")
                    _bn._DIRECT["default"] = None
                    for m in teacher:
                        m.__dict__.pop("_rfn_direct", None)
                    model.__dict__.pop("_mixed_comm", None)
                    self._grad_comm = self._grad_comm2 = None
        if fused_optimizer and model.optimizer_init["class_path"].endswith("AdamW") and \
                next(model.parameters()).is_cuda:
            model.optimizer_init = {**model.optimizer_init,
                                    "init_args": {**model.optimizer_init["init_args"], "fused": True}}
        (opt,), (sch,) = model.configure_optimizers()
        self.optimizer, self.scheduler = opt, sch
        self.fast_step = None
        if fused_optimizer and type(opt) is torch.optim.AdamW and next(model.parameters()).is_cuda:
            from .optim import MultiTensorAdamW
            self.fast_step = MultiTensorAdamW(opt)            # one launch per step; torch's step() where it declines
        groups = model.grad_ready_groups() if hasattr(model, "grad_ready_groups") else None
        self.grads = FlatGradBuffer([p for g in opt.param_groups for p in g["params"]], groups, bucket_mb)
        if getattr(self, "_grad_comm", None) is not None:
            self.grads.use_direct(self._grad_comm, torch.cuda.Stream(device=self._grad_comm.device),
                                  getattr(self, "_grad_comm2", None))
        self.bucket_mb = bucket_mb
        model._optimizer = _OptimizerProxy(self)
        model._grad_buffer = self.grads                      # uda: second buffer for the concurrently running mixed pass
        model._scheduler = sch
        model._backward = self._backward
        self.guard = None
        if dist.is_available() and dist.is_initialized():
            self.broadcast_parameters()
            if dist.get_world_size() > 1:                    # a rank that never arrives must not hang the others for ever
                self.guard = StallGuard(dist.get_rank(), dist.get_world_size())

    def _backward(self, loss, retain_graph=False, last=False):
        """What Lightning's manual_backward does, plus: during the LAST backward pass of a step under data parallelism
        the readiness marks of the MiT stages release finished ranges of the flat gradient buffer to the all-reduce."""
        from . import seg
        # (inside a hipGraph capture of the pass too when the buffer has a communicator of its own -- FlatGradBuffer.use_direct;
        # on_ready declines otherwise and the replayed pass is followed by one reduce of the whole buffer)
        capturing = loss.is_cuda and torch.cuda.is_current_stream_capturing()
        overlap = last and self.data_parallel and os.environ.get("RFN_DDP_OVERLAP", "1") != "0"
        seg._GRAD_READY_CB = self.grads.on_ready if overlap else None
        from . import mfma
        try:
            with mfma.deferred_wgrads():                 # Linear weight gradients: queued, launched per block in groups
                loss.backward(retain_graph=retain_graph)
            if capturing and overlap:
                self.grads.end_capture()
        finally:
            seg._GRAD_READY_CB = None

    def broadcast_parameters(self):
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            dist.broadcast(t.data, src=0)

    def step(self, batch, batch_idx=0, next_batch=None):
        """`next_batch` (optional): the batch of the following step, already on the device -- lets the model compute
        what depends on its inputs only (the frozen ImageNet encoder's features of the next source images) while this
        step's mixed pass runs (uda.prefetch_imnet_features)."""
        if next_batch is not None:
            batch = dict(batch, image_src_next=next_batch["image_src"], semantic_src_next=next_batch.get("semantic_src"),
                         image_trg_next=next_batch.get("image_trg"), image_ref_next=next_batch.get("image_ref"))
        # Python's cyclic collector fires on allocation counts; a step allocates ~10^5 autograd / tensor wrapper objects
        # and a generation-2 pass in the middle of a step stalls the launch thread for ~100 ms (seen as one slow step in
        # ten).  Collect at a step boundary every `gc_interval` steps instead, with the automatic collector off.
        if self.gc_interval:
            import gc
            if self._steps_done == 0:
                gc.collect()
                gc.disable()
            elif self._steps_done % self.gc_interval == 0:
                gc.collect()
            self._steps_done += 1
        if self.guard is not None:
            self.guard.note(f"entering step {self._steps_done}")
        try:
            # (the step's main-stream work on a stream of its own / of another priority: neutral, profiles/r05_main_priority_ab.txt --
            # HIP offers two priority levels here, (0, -1), and the teacher's stream already has the high one)
            self.model.training_step(batch, batch_idx)
            if self.guard is not None:
                self.guard.note(f"step {self._steps_done} queued")
        finally:
            # a crop pre-drawn for a forward that did not happen (exception, mode mismatch) must not leak into the next
            # unrelated extract_crop call
            from . import seg
            seg._PREDRAWN_CROPS.clear()
            seg._DEVICE_CROPS.clear()
        return {k: (float(v) if torch.is_tensor(v) else v) for k, v in self.model.logged.items()} \
            if os.environ.get("RFN_LOG_LOSSES") else None


    def close(self):
        """Give the process its cyclic garbage collector back (step() runs with it disabled between its own collections)."""
        if self.gc_interval and self._steps_done:
            import gc
            gc.enable()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _OptimizerProxy:
    """What `self.optimizers()` returns inside training_step: zero_grad() clears the flat buffer (keeping the views),
    step() all-reduces once and then steps the real optimiser."""

    def __init__(self, trainer):
        self.t = trainer

    def zero_grad(self):
        self.t.grads.zero()

    def step(self):
        self.t.grads.all_reduce_mean(self.t.bucket_mb)        # (and what a concurrently run pass accumulated on the side)
        if self.t.fast_step is not None:
            self.t.fast_step.step()
        else:
            self.t.optimizer.step()

    def __getattr__(self, name):
        return getattr(self.t.optimizer, name)
