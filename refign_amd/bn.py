"""Training-mode BatchNorm2d (+ ReLU) of the decode heads on the hand-written kernels of csrc/bn.hip.

`bn_act_train(x, bn, relu)`: x is an NCHW-SHAPED tensor with channels-last memory (what the convolution kernels hand
over), bn an nn.BatchNorm2d or nn.SyncBatchNorm in training mode; returns relu?(bn(x)) in the same layout, updates bn's
running statistics and num_batches_tracked like the module would, and routes the affine parameters' gradients into the
flat gradient buffer when the trainer provides one.

Data parallelism (the reference trains with `sync_batchnorm: True`; torch/nn/modules/_functions.py:SyncBatchNorm): a
SyncBatchNorm module in an initialised process group of more than one rank gets ONE all-reduce (SUM) of the
(sum x, sum x^2, rows) buffer between the statistics pass and the apply pass, and one of (sum g, sum g xhat) in the
backward; the affine gradients are the local sums, averaged over the ranks with all other gradients by the trainer.  The
collectives are ordinary torch.distributed calls on the module's process group: RCCL calls are capturable, so the passes
stay replayable from hipGraphs (tools/micro/rccl_capture.py).
"""
import torch
import torch.distributed as dist

from . import _lib
from ._tensor import current_stream, on_device, ptr
from .params import grad_sink

_DT16 = {torch.bfloat16: 1, torch.float16: 2}


def data_parallel():
    """True when the step has to behave as one rank of several: a world of more than one process -- or a 1-rank group with
    RFN_DDP_REHEARSAL=1, which makes every exchange of the N > 1 step for real on the one GPU of a development box
    (SyncBatchNorm all-reduces inside the captured student passes, the communicators of the teacher and of the mixed
    pass, the flat gradient all-reduce): everything but the link traffic and the waiting for peers."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("RFN_DDP_REHEARSAL", "0") == "1"


DDP_MODES = ("torch", "direct", "direct3")


def ddp_mode():
    """How one rank of N > 1 exchanges (RFN_DDP_MODE):
      torch    (default) every exchange through torch.distributed's process group -- SyncBatchNorm statistics as all-reduces of
               the default group (the teacher's on a group of its own), the flat gradient buffer in buckets after the passes --
               and eager student passes: one communicator per stream, collectives issued from one host thread in program
               order, the configuration torch's own DDP + SyncBatchNorm runs in.  Nothing here is left to a rehearsal.
      direct   RCCL called directly on the pass's stream (refign_amd/rccl.py): two communicators (student passes on the main
               stream, teacher on the side stream), student passes replayed from hipGraphs with the exchanges inside, passes
               in stream order.
      direct3  + a third communicator, so that the mixed pass runs next to the source pass as on one GPU.
    `direct` / `direct3` have only ever run with ONE rank (RFN_DDP_REHEARSAL=1 on a one-GPU box): concurrent communicators on
    one device can deadlock if two ranks' queues serialise their collectives in different orders, which no one-rank run can
    show -- they stay opt-in until a multi-GPU run has passed tools/ddp_first_contact.sh."""
    import os
    m = os.environ.get("RFN_DDP_MODE", "torch")
    if m not in DDP_MODES:
        raise RuntimeError(f"RFN_DDP_MODE={m!r}: one of {DDP_MODES}")
    return m


# RCCL called directly on the calling stream for the STUDENT's exchanges (refign_amd/rccl.py; set up by the trainer): `default` is the
# communicator of the pass on the main stream, `current` the one of the pass being captured / run inside direct_comm().
_DIRECT = {"default": None, "current": None}


def direct_comm(comm):
    """Context manager: the student's statistics exchanges inside it go through `comm` (a rccl.DirectComm) -- a pass that
    runs NEXT TO another pass needs a communicator of its own (uda._mixed_capture_context)."""
    import contextlib

    @contextlib.contextmanager
    def ctx():
        saved, _DIRECT["current"] = _DIRECT["current"], comm
        try:
            yield
        finally:
            _DIRECT["current"] = saved
    return ctx()


def _exchange_comm(bn):
    """The rccl.DirectComm a module's exchange goes through, None = torch's process group.  Modules that run on a
    stream of their own (the teacher's) carry their communicator (`_rfn_direct`, set by the trainer)."""
    own = getattr(bn, "_rfn_direct", None)
    if own is not None:
        return own
    if getattr(bn, "process_group", None) is not None:
        return None
    return _DIRECT["current"] if _DIRECT["current"] is not None else _DIRECT["default"]


def _all_reduce(buf, group, comm):
    if comm is not None:
        comm.all_reduce_(buf)
    else:
        dist.all_reduce(buf, group=group)


def sync_group(bn):
    """The process group a module's statistics are exchanged over: None for plain BatchNorm2d and for a world of one
    (torch's SyncBatchNorm also normalises locally then), else the module's group (default: the world)."""
    if not isinstance(bn, torch.nn.SyncBatchNorm) or not data_parallel():
        return None
    return bn.process_group if bn.process_group is not None else dist.group.WORLD


def usable(x, bn, dtype, channels=None, count=None):
    """`x`: the tensor BatchNorm is applied to -- or, with `channels`, the INPUT of the convolution in front of it (the caller
    asks before it convolves; what has to be a multiple of 8 is the normalised tensor's channel count, not the convolution's
    input: round 4 asked about the wrong one, and layers with 1 or 84 input channels fell to the library's BatchNorm).
    `count`: values per channel of the NORMALISED tensor (the convolution's output: batch x OH x OW) when x is the input --
    one value per channel has no variance, and torch raises there ('Expected more than 1 value per channel'); so do we,
    by leaving the call to torch."""
    c = x.shape[1] if channels is None else channels
    n = x.shape[0] * x.shape[2] * x.shape[3] if count is None else count
    return (x.is_cuda and dtype in _DT16 and type(bn) in (torch.nn.BatchNorm2d, torch.nn.SyncBatchNorm) and bn.training
            and bn.affine and bn.track_running_stats and bn.momentum is not None and c % 8 == 0 and c == bn.num_features
            and n > 1)


# The four kernel passes (csrc/bn.hip) on (B, H, W, C) 16-bit tensors; module-level so that tests/test_ddp_cpu.py can put
# CPU restatements in their place and drive the exchange logic below over gloo.
def _stats_fwd(xh, sums):
    T, C = xh.numel() // xh.shape[-1], xh.shape[-1]
    with on_device(xh.device):
        _lib.check(_lib.load_library().rfn_bn_stats_fwd(ptr(xh), ptr(sums), T, C, _DT16[xh.dtype], current_stream(xh.device)),
                   "bn_stats_fwd")


def _apply_fwd(xh, weight, bias, y, sums, bn, relu):
    T, C = xh.numel() // xh.shape[-1], xh.shape[-1]
    if y.stride(-2) != C:                                  # a channel slice of a wider channels-last tensor
        with on_device(xh.device):
            _lib.check(_lib.load_library().rfn_bn_apply_fwd_ld(ptr(xh), ptr(weight), ptr(bias), ptr(y), y.stride(-2), ptr(sums),
                                                               ptr(bn.running_mean), ptr(bn.running_var), T, C,
                                                               float(bn.eps), float(bn.momentum), int(relu),
                                                               _DT16[xh.dtype], current_stream(xh.device)), "bn_apply_fwd_ld")
        return
    with on_device(xh.device):
        _lib.check(_lib.load_library().rfn_bn_apply_fwd(ptr(xh), ptr(weight), ptr(bias), ptr(y), ptr(sums),
                                                        ptr(bn.running_mean), ptr(bn.running_var), T, C, float(bn.eps),
                                                        float(bn.momentum), int(relu), _DT16[xh.dtype],
                                                        current_stream(xh.device)), "bn_apply_fwd")


def _stats_bwd(xh, gy, sums, weight, bias, bsums, eps, relu):
    T, C = xh.numel() // xh.shape[-1], xh.shape[-1]
    with on_device(xh.device):
        _lib.check(_lib.load_library().rfn_bn_stats_bwd(ptr(xh), ptr(gy), ptr(sums), ptr(weight), ptr(bias), ptr(bsums), T, C,
                                                        eps, int(relu), _DT16[xh.dtype], current_stream(xh.device)),
                   "bn_stats_bwd")


def _apply_bwd(xh, gy, sums, bsums, weight, bias, gx, eps, relu):
    T, C = xh.numel() // xh.shape[-1], xh.shape[-1]
    with on_device(xh.device):
        _lib.check(_lib.load_library().rfn_bn_apply_bwd(ptr(xh), ptr(gy), ptr(sums), ptr(bsums), ptr(weight), ptr(bias),
                                                        ptr(gx), T, C, eps, int(relu), _DT16[xh.dtype],
                                                        current_stream(xh.device)), "bn_apply_bwd")


class _BNActTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xh, weight, bias, bn, relu, group, out=None, sums=None):
        # xh: (B, H, W, C) contiguous, 16-bit; out (gradient-free callers): where the result goes -- (B, H, W, C) with the
        # strides of a channel slice of a contiguous (B, H, W, C') tensor
        C = xh.shape[-1]
        y = torch.empty_like(xh) if out is None else out
        if sums is None:                                    # (else: the producer of xh left them, csrc/dwconv.hip STATS)
            sums = torch.empty(2 * C + 1, dtype=torch.float64, device=xh.device)     # fp64: csrc/bn.hip header
            _stats_fwd(xh, sums)
        comm = _exchange_comm(bn) if group is not None else None
        if group is not None:
            _all_reduce(sums, group, comm)
        _apply_fwd(xh, weight, bias, y, sums, bn, relu)
        bn.num_batches_tracked.add_(1)
        if any(ctx.needs_input_grad[:3]):
            ctx.save_for_backward(xh, sums, weight, bias)
            ctx.eps, ctx.relu, ctx.group, ctx.comm = float(bn.eps), relu, group, comm
        return y

    @staticmethod
    def backward(ctx, gy):
        xh, sums, weight, bias = ctx.saved_tensors
        C = xh.shape[-1]
        if gy.dtype != xh.dtype:
            gy = gy.to(xh.dtype)
        if not gy.is_contiguous():
            gy = gy.contiguous()
        gx = torch.empty_like(xh)
        bsums = torch.empty((2, C), dtype=torch.float32, device=xh.device)
        _stats_bwd(xh, gy, sums, weight, bias, bsums, ctx.eps, ctx.relu)
        local = bsums
        if ctx.group is not None:                           # parameter gradients: this replica's sums
            local = bsums.clone()
            _all_reduce(bsums, ctx.group, ctx.comm)
        _apply_bwd(xh, gy, sums, bsums, weight, bias, gx, ctx.eps, ctx.relu)
        gw = gb = None
        if ctx.needs_input_grad[1]:
            sink = grad_sink(weight)
            if sink is not None:
                sink.add_(local[1])
            else:
                gw = local[1].to(weight.dtype)
        if ctx.needs_input_grad[2]:
            sink = grad_sink(bias)
            if sink is not None:
                sink.add_(local[0])
            else:
                gb = local[0].to(bias.dtype)
        return (gx if ctx.needs_input_grad[0] else None), gw, gb, None, None, None, None, None


def slice_out_ok(out, xshape):
    """`out` can receive a (B, C, H, W)-shaped result directly: an NCHW-shaped view whose memory is a channel slice of a
    contiguous channels-last tensor (16-byte aligned, pitch a multiple of 8)."""
    B, C, H, W = xshape
    if tuple(out.shape) != (B, C, H, W) or out.stride(1) != 1 or out.stride(3) % 8 or out.stride(3) < C:
        return False
    p = out.stride(3)
    return out.stride(2) == W * p and out.stride(0) == H * W * p and out.data_ptr() % 16 == 0


def bn_act_train(x, bn, relu, dtype, out=None, sums=None):
    """act(bn(x)) with batch statistics, `relu`: False / 0 none, True / 1 ReLU, 3 LeakyReLU(0.1) (the activation codes of
    the GEMM entry points); x NCHW-shaped (any memory format, converted to channels-last 16-bit if it is
    not already); returns an NCHW-shaped channels-last tensor."""
    xh = x.permute(0, 2, 3, 1)
    if xh.dtype != dtype:
        xh = xh.to(dtype)
    if not xh.is_contiguous():
        xh = xh.contiguous()
    w = bn.weight if bn.weight.dtype == torch.float32 else bn.weight.float()
    b = bn.bias if bn.bias.dtype == torch.float32 else bn.bias.float()
    if out is not None:
        if torch.is_grad_enabled() and (xh.requires_grad or w.requires_grad) or out.dtype != dtype \
                or not slice_out_ok(out, x.shape):
            raise RuntimeError("bn_act_train(out=...): gradient-free calls into a channels-last channel slice only")
        _BNActTrain.apply(xh, w, b, bn, relu, sync_group(bn), out.permute(0, 2, 3, 1), sums)
        return out
    return _BNActTrain.apply(xh, w, b, bn, relu, sync_group(bn), None, sums).permute(0, 3, 1, 2)
