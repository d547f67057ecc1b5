"""Weight-gradient kernels off the critical path of the backward pass.

The backward of a layer has two independent halves: the INPUT gradient, which the next layer down is waiting for, and
the PARAMETER gradients, which nothing needs before the optimizer step.  The student's backward is a chain of small
kernels (a few hundred workgroups each on a 256-CU chip), so the parameter-gradient kernels (split-T weight-gradient
GEMMs, depthwise weight gradients) are launched on a second stream: they fill idle CUs next to the input-gradient
chain instead of lengthening it.  Ordering: the side stream waits for an event recorded after the producer of the
incoming gradient; `join()` -- queued as an autograd-engine callback, i.e. run at the end of the backward pass that
forked -- makes the main stream wait for the side stream, so every gradient is complete before the next backward pass, the all-reduce and the
optimizer step.  Parameter gradients are accumulated with fp32 atomics or by one writer per parameter on this one side
stream, so there is no write conflict with the main stream.  The same code is captured into the student-pass hipGraphs
(the event wait pulls the side stream into the capture; the join closes the fork).

MEASURED AND OFF BY DEFAULT (RFN_WGRAD_STREAM=1 enables it): captured into the student-pass hipGraphs the ~2 000 fork /
join edges per step make the replay far slower -- 449 ms per step against 244 ms on one stream (MI355X, ROCm 7.2: a graph
with cross-stream branches is replayed through internal streams with a synchronisation per edge).  Kept as the knob that
documents the experiment; the single-stream order is the product path.
"""
import os

import torch

_ENABLED = os.environ.get("RFN_WGRAD_STREAM", "0") == "1"
_streams = {}
_dirty = set()


def _side(device):
    s = _streams.get(device)
    if s is None:
        s = _streams[device] = torch.cuda.Stream(device=device)
    return s


def fork(device, fn, *tensors):
    """Run fn() on the side stream of `device`, after everything enqueued so far on the current stream; `tensors` are
    the arguments it reads (kept alive for the side stream)."""
    if not (_ENABLED and device.type == "cuda"):
        return fn()
    cur = torch.cuda.current_stream(device)
    side = _side(device)
    ev = torch.cuda.Event()
    ev.record(cur)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        out = fn()
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    if not _dirty:
        # close the fork when the running backward pass ends, whoever started it (loss.backward(), autograd.grad, ...)
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join)
        except RuntimeError:
            pass                                   # not inside a backward pass: the caller joins
    _dirty.add(device)
    return out


def join():
    """The current stream of every device with forked work waits for its side stream."""
    for device in list(_dirty):
        torch.cuda.current_stream(device).wait_stream(_side(device))
    _dirty.clear()
