"""hipGraph replay of the gradient-free, shape-static parts of the training step.

One Refign step issues ~14 000 kernel launches and the host needs ~25 us for each (Python module tree, dispatcher,
library heuristics): at 1080x1920 the host, not the GPU, is the critical resource (profiles/r01_*: 335 ms of host
time for 323 ms of kernel time).  The EMA-teacher forward, align, refine and the ImageNet feature extractor have no
autograd, no host-side control flow that depends on data, and the same shapes every step -- their ~5 000 launches are
captured once (after eager warm-up calls that populate every cache) and replayed with one host call.

Safety rules:
  * eager fallback, permanently, if capture throws (e.g. a library call that is not capturable) -- recoverable, see
    _capture;
  * a capture is keyed on input shapes/dtypes/devices, the autocast state and a module "generation" that the owner
    bumps whenever cached derived tensors may have been re-allocated (train()/eval(), load_state_dict, .to());
  * parameters are only ever updated IN PLACE between replays (optimizer, EMA, params.refresh) -- same addresses;
  * outputs are static buffers: valid until the next call.
  * no collectives inside a captured region: the teacher's decode head (SyncBatchNorm in train mode under DDP, D9) stays
    eager between the captured teacher backbone and the captured align + refine.
RFN_HIP_GRAPH=0 disables (pure eager).  On by default for every world size: the captured regions contain no collective,
the capture runs in thread-local error mode (the RCCL watchdog thread cannot invalidate it), and a capture that fails
anyway is recoverable -- the thread is put back on its original stream and the region runs eagerly from then on
(tests/test_step_gpu.py::test_failed_graph_capture_falls_back_to_eager).  Exercised on the GPU with a 1-rank RCCL
process group (same module tree as N > 1: SyncBatchNorm everywhere); not with several ranks (one-GPU boxes).
"""
import contextlib
import gc
import os
import warnings

import torch


def enabled():
    return os.environ.get("RFN_HIP_GRAPH", "1") != "0"


def concurrent_stream(device, peers, priority=0, tries=8, spin_cycles=4_000_000):
    """A new stream whose kernels really run NEXT TO those of `peers`.  The runtime multiplexes a process's streams onto a few
    hardware queues (4 by default) in creation order, and two streams that land on ONE queue do not overlap -- their kernels
    and graph replays run back to back.  Which queue a new stream gets depends on everything created before it (capture
    streams, RCCL's streams, the streams of an earlier model), so a fixed recipe breaks when the step's history changes:
    round 6 found the mixed pass of the `adapt_to_ref` configuration (one more capture before the stream is made) sharing the
    main stream's queue -- every pass at its stand-alone duration, one after the other (profiles/r06_stream_priority_ab.txt:
    121.9 ms per step, 106.9 with probed streams).
    So the stream is PROBED: a ~1.7 ms spin kernel on each peer and on the candidate, timed with events; a candidate that
    serialises with a peer is kept alive (so that the next one lands elsewhere) and the next one is tried.  Costs ~10-30 ms, once per stream.
    Returns (stream, report)."""
    cur = torch.cuda.current_stream(device)

    def span(streams):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record(cur)
        for st in streams:
            if st != cur:
                st.wait_event(start)
            with torch.cuda.stream(st):
                torch.cuda._sleep(spin_cycles)
        for st in streams:
            if st != cur:
                cur.wait_stream(st)
        end.record(cur)
        end.synchronize()
        return start.elapsed_time(end)

    peers = [p for p in peers if p is not None]
    span(peers[:1] or [cur])
    one = min(span([p]) for p in (peers or [cur]))
    rejected, report = [], []
    for _ in range(tries):
        cand = torch.cuda.Stream(device=device, priority=priority)
        span([cand])
        worst = max([span([p, cand]) for p in peers] or [one])
        report.append(round(worst / one, 2))
        if worst < 1.5 * one:
            break
        rejected.append(cand)
    else:
        warnings.warn(f"refign_amd: no stream found that overlaps with its peers (slow-down factors {report}); "
                      "GPU_MAX_HW_QUEUES may be too small")
        cand = rejected[0]
    _KEPT_STREAMS.extend(rejected)
    return cand, report


_KEPT_STREAMS = []


@contextlib.contextmanager
def _no_cyclic_gc():
    """No cyclic garbage collection while a capture is open: a collection that happens to run inside the captured region
    finalises whatever cyclic garbage exists -- graphs, events, tensors with recorded streams of an earlier model -- and
    some of their destructors make runtime calls that are illegal during a capture; the process aborts (seen in round 3 in
    the test suite, on the ninth model of a process).  torch.cuda.graph collects BEFORE it begins; reference counting
    keeps freeing everything that is not a cycle."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def _fresh_containers(out):
    """Static output tensors are handed out as they are; python containers around them are rebuilt per call (callers
    such as the HRDA head rescale the crop-box list in place)."""
    if isinstance(out, (list, tuple)):
        return type(out)(_fresh_containers(o) for o in out)
    return out


class GraphedNoGrad:
    def __init__(self, fn, name, warmup=1):
        self.fn, self.name, self.warmup = fn, name, warmup
        self.generation = 0
        self.states = {}

    def reset(self):
        """Drop every capture (cached derived tensors may have moved)."""
        self.generation += 1
        self.states.clear()

    def __call__(self, *tensors):
        if not tensors[0].is_cuda or not enabled() or torch.is_grad_enabled() or \
                torch.cuda.is_current_stream_capturing():          # inside an enclosing capture: recorded inline
            return self.fn(*tensors)
        key = (tuple((tuple(t.shape), t.dtype, t.device) for t in tensors), torch.is_autocast_enabled("cuda"),
               torch.get_autocast_dtype("cuda"), self.generation)
        st = self.states.get(key)
        if st is None:
            st = self.states[key] = {"calls": 0, "graph": None, "failed": False}
        if st["failed"]:
            return self.fn(*tensors)
        if st["graph"] is None:
            st["calls"] += 1
            if st["calls"] <= self.warmup:
                return self.fn(*tensors)
            try:
                self._capture(st, tensors)
            except Exception as e:                          # not capturable here: stay eager for good
                st["failed"] = True
                warnings.warn(f"refign_amd.graphs: capture of '{self.name}' failed ({type(e).__name__}: {e}); "
                              f"running it eagerly")
                torch.cuda.synchronize()
                return self.fn(*tensors)
        for s, t in zip(st["inputs"], tensors):
            s.copy_(t)
        st["graph"].replay()
        return _fresh_containers(st["outputs"])

    def _capture(self, st, tensors):
        inputs = [t.clone() for t in tensors]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self.fn(*inputs)                                # once more on the side stream (stream-local workspaces)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread_local: calls made by other host threads during the capture (the RCCL watchdog of torch.distributed
        # polls its events) must not invalidate it
        try:
            with _no_cyclic_gc(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                outputs = self.fn(*inputs)
        except BaseException:
            # torch.cuda.graph.__exit__ raises from capture_end() BEFORE it restores the stream: the thread would stay
            # on the (now invalidated) capture stream and every later launch would fail.  Put it back.
            torch.cuda.set_stream(cur)
            raise
        st["graph"], st["inputs"], st["outputs"] = g, inputs, outputs


class GraphedStep:
    """hipGraph replay of a whole forward + backward pass of the student (source pass, mixed pass).

    `fn(*tensors)` runs forward, loss and `backward()` and returns a tuple of loss tensors; its kernels -- including the
    ones the autograd engine launches -- are recorded once and replayed with one host call per step afterwards.  What
    makes that legal here: shapes never change; every per-step host decision of the pass is device data (the HRDA crop
    offsets, seg.DeviceBox; stochastic depth / dropout masks come from the device generator, which torch re-seeds per
    replay); parameter gradients accumulate IN PLACE into the flat gradient buffer (trainer.FlatGradBuffer) and
    parameters / cached 16-bit copies are only updated in place between replays; BatchNorm statistics are in-place
    buffer updates.  Under data parallelism the pass contains the SyncBatchNorm statistics exchanges of the decode heads
    (refign_amd/bn.py: RCCL all-reduces, which are capturable -- tools/micro/rccl_capture.py); replaying them has only been
    run with a 1-rank group here (one-GPU development boxes: RFN_DDP_REHEARSAL=1 makes a 1-rank group do everything one
    rank of N does).  Every rank captures on the same call, a capture records but does not execute its collectives, and a
    rank whose capture throws runs the same collectives eagerly, so the ranks stay matched either way.  Under data
    parallelism the capture is ON when the exchanges are RCCL calls of our own on the pass's stream (refign_amd/rccl.py:
    plain kernel nodes, and the mixed pass runs next to the source pass on a communicator of its own -- rehearsal
    192.3 ms/step: RFN_DDP_MODE=direct / direct3) and OFF when they go through torch's process group (RFN_DDP_MODE=torch, the
    default for N > 1: a captured collective is then a cross-stream branch of the graph; 216.4 ms graphed, 214-231 ms eager,
    no gain) -- see usable().
    The first `warmup` calls run eagerly (they create every lazily cached constant / derived tensor); a capture that
    throws leaves the pass eager for good, like GraphedNoGrad."""

    def __init__(self, fn, name, warmup=2, shared=None, capture_context=None, after_capture=None, on_replay=None):
        """`shared`: a dict the passes of ONE model share -- they are never live at the same time, so their graphs
        capture into one memory pool (held there; it dies with the model's graphs, never outlives them).  A pass that
        may run NEXT TO another one gets a pool of its own (shared=None).
        `capture_context`: callable returning a context manager the capture runs inside (e.g. the gradient buffer the
        captured backward kernels are to accumulate into).
        `after_capture()` -> anything, kept with the graph; `on_replay(that)` is called after every replay of it (the ranges of
        the gradient buffer whose all-reduce is part of the graph: trainer.FlatGradBuffer.end_capture / replayed)."""
        self.fn, self.name, self.warmup = fn, name, warmup
        self.capture_context = capture_context
        self.after_capture, self.on_replay = after_capture, on_replay
        self._last = None
        self.shared = {} if shared is None else shared
        self.generation = 0
        self.states = {}

    def reset(self):
        self.generation += 1
        self.states.clear()
        self._last = None
        self.shared.pop("pool", None)

    def captured(self):
        """True once the most recently used input signature replays from a graph."""
        return self._last is not None and self._last["graph"] is not None

    @staticmethod
    def usable(t):
        if not (t.is_cuda and enabled() and os.environ.get("RFN_GRAPH_STUDENT", "1") != "0"):
            return False
        import torch.distributed as dist
        from .bn import data_parallel
        if data_parallel():
            # on when the statistics exchanges are RCCL calls of our own on the pass's stream (plain kernel nodes:
            # RFN_DDP_MODE=direct / direct3), off when they go through torch's process group (cross-stream branches of the
            # graph: no gain over eager, measured)
            from . import bn
            return bn.ddp_mode() != "torch" and bn._DIRECT["default"] is not None and dist.get_backend() == "nccl"
        return True

    def __call__(self, *tensors):
        if not self.usable(tensors[0]):
            return self.fn(*tensors)
        key = (tuple((tuple(t.shape), t.dtype, t.device) for t in tensors), torch.is_autocast_enabled("cuda"),
               torch.get_autocast_dtype("cuda"), self.generation)
        st = self.states.get(key)
        if st is None:
            st = self.states[key] = {"calls": 0, "graph": None, "failed": False}
        self._last = st
        if st["failed"]:
            return self.fn(*tensors)
        if st["graph"] is None:
            st["calls"] += 1
            if st["calls"] <= self.warmup:
                return self.fn(*tensors)
            try:
                self._capture(st, tensors)
            except Exception as e:
                st["failed"] = True
                warnings.warn(f"refign_amd.graphs: capture of '{self.name}' failed ({type(e).__name__}: {e}); "
                              f"running it eagerly")
                torch.cuda.synchronize()
                return self.fn(*tensors)
        for s, t in zip(st["inputs"], tensors):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t)
        st["graph"].replay()
        if self.on_replay is not None:
            self.on_replay(st.get("extra"))
        # copies, not the pool tensors themselves: the passes of a model share one pool and are not always replayed in the
        # order they were captured in (a second input signature of the source pass is captured AFTER the mixed pass and
        # replayed before it) -- a later replay of the other pass may then reuse the blocks these few scalars live in
        return tuple(o.clone() for o in st["outputs"])

    def _capture(self, st, tensors):
        inputs = [t.clone() for t in tensors]
        cur = torch.cuda.current_stream()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        import contextlib
        ctx = self.capture_context() if self.capture_context is not None else contextlib.nullcontext()
        try:
            with ctx, _no_cyclic_gc():
                with torch.cuda.graph(g, pool=self.shared.get("pool"), capture_error_mode="thread_local"):
                    outputs = self.fn(*inputs)
        except BaseException:
            torch.cuda.set_stream(cur)
            raise
        self.shared.setdefault("pool", g.pool())
        st["graph"], st["inputs"], st["outputs"] = g, inputs, outputs
        st["extra"] = self.after_capture() if self.after_capture is not None else None


class GraphedSplitStep(GraphedStep):
    """A student pass as TWO replayable units sharing one memory pool: `forward(*tensors)` (network forward up to the
    low-resolution class logits; the autograd graph of the captured call is kept) and `backward(*tensors)` (losses +
    `backward()` through that graph) -- the layout of torch's make_graphed_callables, driven by hand.

    Why two units: only the LOSS of the mixed pass depends on the teacher's pseudo-labels (segmentation_model.py:214-250: the
    mixed image is a function of the source / target images and the source labels alone), so its forward can be queued at the
    start of the step and run next to the teacher branch; and an event between the source pass's two units orders the decode
    head's BatchNorm running-statistics updates of the two forwards (source first, as in the reference) without waiting for
    the source backward.

    `fwd_fn(*tensors) -> held` (anything; tensors inside keep their grad_fn), `bwd_fn(held, *tensors) -> tuple of losses`.
    Every call of forward() must be followed by exactly one backward() before the next forward().  Warm-up calls and a
    failed capture run both functions eagerly, like GraphedStep."""

    def __init__(self, fwd_fn, bwd_fn, name, forward_state=None, agree=None, **kw):
        """`forward_state()` -> the tensors the forward updates in place as a SIDE EFFECT (BatchNorm running statistics and
        batch counters): kept as they are across the one eager re-run of the forward that a failed backward capture needs.
        `agree(failed) -> failed on ANY rank` (data parallelism, called once, on the step the backward is captured): every
        rank then takes the same way out, so the statistics exchanges of the re-run forward are issued on all ranks or on none."""
        super().__init__(None, name, **kw)
        self.fwd_fn, self.bwd_fn = fwd_fn, bwd_fn
        self.forward_state, self.agree = forward_state, agree
        self._held = None            # (state, held, fwd inputs) of the forward that awaits its backward

    def reset(self):
        super().reset()
        self._held = None

    def _ctx(self):
        import contextlib
        return self.capture_context() if self.capture_context is not None else contextlib.nullcontext()

    def forward(self, *tensors, variant=None):
        if not self.usable(tensors[0]):
            self._held = (None, self.fwd_fn(*tensors), tensors)
            return
        key = (tuple((tuple(t.shape), t.dtype, t.device) for t in tensors), torch.is_autocast_enabled("cuda"),
               torch.get_autocast_dtype("cuda"), self.generation, variant)
        st = self.states.get(key)
        if st is None:
            st = self.states[key] = {"calls": 0, "graph": None, "graph_bwd": None, "failed": False}
        self._last = st
        if st["failed"]:
            self._held = (None, self.fwd_fn(*tensors), tensors)
            return
        if st["graph"] is None:
            st["calls"] += 1
            if st["calls"] <= self.warmup:
                self._held = (None, self.fwd_fn(*tensors), tensors)
                return
            try:
                inputs = [t.clone() for t in tensors]
                cur = torch.cuda.current_stream()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                try:
                    with self._ctx(), _no_cyclic_gc():
                        with torch.cuda.graph(g, pool=self.shared.get("pool"), capture_error_mode="thread_local"):
                            held = self.fwd_fn(*inputs)
                except BaseException:
                    torch.cuda.set_stream(cur)
                    raise
                self.shared.setdefault("pool", g.pool())
                st["graph"], st["inputs"], st["held"] = g, inputs, held
                # the static inputs already hold this call's values (they were cloned from them), and the captured autograd graph
                # may have saved them (a single-scale backbone's first convolution saves the image for its weight gradient): an
                # in-place copy before the backward capture would bump their version under it
                st["graph"].replay()
                self._held = (st, None, tensors)
                return
            except Exception as e:
                st["failed"] = True
                warnings.warn(f"refign_amd.graphs: capture of '{self.name}' (forward) failed ({type(e).__name__}: {e}); "
                              f"running it eagerly")
                torch.cuda.synchronize()
                self._held = (None, self.fwd_fn(*tensors), tensors)
                return
        for s, t in zip(st["inputs"], tensors):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t)
        st["graph"].replay()
        self._held = (st, None, tensors)

    def backward(self, *tensors):
        st, held, fwd_tensors = self._held
        self._held = None
        if st is None:
            return self.bwd_fn(held, *tensors)
        if st["graph_bwd"] is None:
            try:
                inputs = [t.clone() for t in tensors]
                cur = torch.cuda.current_stream()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                try:
                    with self._ctx(), _no_cyclic_gc():
                        with torch.cuda.graph(g, pool=self.shared.get("pool"), capture_error_mode="thread_local"):
                            outputs = self.bwd_fn(st["held"], *inputs)
                except BaseException:
                    torch.cuda.set_stream(cur)
                    raise
                st["graph_bwd"], st["inputs_bwd"], st["outputs"] = g, inputs, outputs
                st["held"] = None                 # the autograd graph has been consumed; its buffers live in the pool
                st["extra"] = self.after_capture() if self.after_capture is not None else None
                failure = None
            except Exception as e:
                failure = e
            failed = failure is not None
            if self.agree is not None:
                failed = bool(self.agree(failed))        # one tiny collective, on this step only: all ranks go the same way
            if failed:
                # The forward of this step has only run as a replay, and the failed capture may have consumed its autograd graph:
                # the pass is run again eagerly, and is eager for good afterwards.  The replayed forward has ALREADY made the
                # forward's in-place side effects (BatchNorm running statistics, num_batches_tracked): the re-run's second
                # update is undone from a snapshot, so the step leaves the state a single forward leaves.  Its statistics
                # exchanges (data parallelism) are issued a second time by EVERY rank (agree), never by one rank alone.  The
                # device generator advances twice (drop-path masks of the re-run differ from the replay's: the re-run's
                # forward and backward are consistent with each other, which is what the gradients need).
                st["failed"], st["graph"], st["held"], st["graph_bwd"] = True, None, None, None
                warnings.warn(f"refign_amd.graphs: capture of '{self.name}' (backward) failed "
                              f"({type(failure).__name__ if failure is not None else 'on another rank'}: {failure}); "
                              f"running it eagerly")
                torch.cuda.synchronize()
                keep = [(t, t.clone()) for t in (self.forward_state() if self.forward_state is not None else [])]
                out = self.bwd_fn(self.fwd_fn(*fwd_tensors), *tensors)
                with torch.no_grad():                    # (after the backward: autograd has saved the running buffers)
                    for t, saved in keep:
                        t.copy_(saved)
                return out
        for s, t in zip(st["inputs_bwd"], tensors):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t)
        st["graph_bwd"].replay()
        if self.on_replay is not None:
            self.on_replay(st.get("extra"))
        return tuple(o.clone() for o in st["outputs"])

    def captured(self):
        return self._last is not None and self._last["graph"] is not None and self._last.get("graph_bwd") is not None

    def any_failed(self):
        return any(st["failed"] for st in self.states.values())

    def give_up(self):
        """Eager from now on, for every input signature seen so far and for new ones of this generation (the owner calls this on
        the passes of a model when ONE of them could not be captured: the passes then all run eagerly, the one combination every
        configuration is tested in -- a replayed pass next to an eagerly run one is not)."""
        if self._held is not None and self._held[0] is not None:
            raise RuntimeError("GraphedSplitStep.give_up between forward() and backward()")
        for st in self.states.values():
            st["failed"], st["graph"], st["graph_bwd"], st["held"] = True, None, None, None
        self.warmup = 1 << 60

    def __call__(self, *a, **k):
        raise TypeError("GraphedSplitStep: call forward() and backward()")


def _leaves(out, acc):
    """the tensors of a nested tuple / list structure, in order"""
    if torch.is_tensor(out):
        acc.append(out)
    elif isinstance(out, (list, tuple)):
        for o in out:
            _leaves(o, acc)
    return acc


def _rebuild(out, it):
    if torch.is_tensor(out):
        return next(it)
    if isinstance(out, (list, tuple)):
        return type(out)(_rebuild(o, it) for o in out)
    return out


class _SegmentSplice(torch.autograd.Function):
    """The captured sub-network as ONE node of the surrounding (eager) autograd graph: forward hands out the static outputs of the
    replayed forward graph, backward copies the incoming gradients into the static gradient buffers and replays the backward
    graph (which accumulates the parameters' gradients where the captured kernels put them: the flat gradient buffer)."""

    @staticmethod
    def forward(ctx, st, _anchor, *outs):
        ctx.st = st
        return tuple(o.view_as(o) for o in outs)

    @staticmethod
    def backward(ctx, *grads):
        st = ctx.st
        for buf, g in zip(st["grads"], grads):
            if g is None:
                buf.zero_()
            else:
                buf.copy_(g)
        st["graph_bwd"].replay()
        return (None, None) + (None,) * len(grads)


class GraphedSegment:
    """A collective-free PART of a student pass (the MiT backbone: 52 blocks, ~90 % of the pass's launches) as a forward and a
    backward hipGraph spliced into an otherwise eager pass -- the form for data parallelism through torch.distributed
    (RFN_DDP_MODE=torch, the N > 1 default): the decode heads' SyncBatchNorm exchanges stay ordinary torch.distributed calls, issued
    eagerly from one host thread in program order, BETWEEN the replays (graph -> collectives -> graph), and the launch-bound bulk of
    the pass stops costing host time (round 5: eager student passes alone cost 12.8 ms of the step).  The layout of torch's
    make_graphed_callables, driven by hand because the captured kernels accumulate parameter gradients in place (no gradient
    tensors to hand back).  `fn(*tensors)` -> nested tuples / lists of tensors (other objects pass through, e.g. the crop box).
    Warm-up calls and a failed capture run `fn` eagerly; every call's forward must be followed by its backward before the next call."""

    def __init__(self, fn, name, warmup=2):
        self.fn, self.name, self.warmup = fn, name, warmup
        self.generation = 0
        self.states = {}
        self._anchor = None

    def reset(self):
        self.generation += 1
        self.states.clear()

    def captured(self):
        return any(st["graph"] is not None for st in self.states.values())

    def __call__(self, *tensors):
        if not (tensors[0].is_cuda and enabled() and torch.is_grad_enabled()) or torch.cuda.is_current_stream_capturing():
            return self.fn(*tensors)
        key = (tuple((tuple(t.shape), t.dtype, t.device) for t in tensors), torch.is_autocast_enabled("cuda"),
               torch.get_autocast_dtype("cuda"), self.generation)
        st = self.states.get(key)
        if st is None:
            st = self.states[key] = {"calls": 0, "graph": None, "failed": False}
        if st["failed"]:
            return self.fn(*tensors)
        if st["graph"] is None:
            st["calls"] += 1
            if st["calls"] <= self.warmup:
                return self.fn(*tensors)
            try:
                self._capture(st, tensors)
            except Exception as e:
                st["failed"], st["graph"] = True, None
                warnings.warn(f"refign_amd.graphs: capture of '{self.name}' failed ({type(e).__name__}: {e}); running it eagerly")
                torch.cuda.synchronize()
                return self.fn(*tensors)
        for s_, t in zip(st["inputs"], tensors):
            if s_.data_ptr() != t.data_ptr():
                s_.copy_(t)
        st["graph"].replay()
        if self._anchor is None or self._anchor.device != tensors[0].device:
            self._anchor = torch.zeros(1, device=tensors[0].device, requires_grad=True)
        outs = _SegmentSplice.apply(st, self._anchor, *st["outs"])
        return _rebuild(st["structure"], iter(outs))

    def _capture(self, st, tensors):
        inputs = [t.clone() for t in tensors]
        cur = torch.cuda.current_stream()
        torch.cuda.synchronize()
        fwd, bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        try:
            with _no_cyclic_gc():
                with torch.cuda.graph(fwd, capture_error_mode="thread_local"):
                    structure = self.fn(*inputs)
                outs = [o for o in _leaves(structure, []) if o.requires_grad]
                if len(outs) != len(_leaves(structure, [])):
                    raise RuntimeError("every tensor a graphed segment returns must carry a gradient")
                grads = [torch.empty_like(o) for o in outs]
                with torch.cuda.graph(bwd, pool=fwd.pool(), capture_error_mode="thread_local"):
                    torch.autograd.backward(outs, grad_tensors=grads)
        except BaseException:
            torch.cuda.set_stream(cur)
            raise
        st["graph"], st["graph_bwd"], st["inputs"] = fwd, bwd, inputs
        st["outs"], st["grads"] = [o.detach() for o in outs], grads
        st["structure"] = structure
