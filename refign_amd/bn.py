"""Training-mode BatchNorm2d (+ ReLU) of the decode heads on the hand-written kernels of csrc/bn.hip.

`bn_act_train(x, bn, relu)`: x is an NCHW-SHAPED tensor with channels-last memory (what the convolution kernels hand
over), bn a plain nn.BatchNorm2d in training mode; returns relu?(bn(x)) in the same layout, updates bn's running
statistics and num_batches_tracked like the module would, and routes the affine parameters' gradients into the flat
gradient buffer when the trainer provides one.  SyncBatchNorm modules are NOT taken here (the caller keeps torch's
implementation for them): their cross-rank statistics exchange is untested on this one-GPU development setup.
"""
import torch

from . import _lib
from ._tensor import current_stream, on_device, ptr
from .params import grad_sink

_DT16 = {torch.bfloat16: 1, torch.float16: 2}


def usable(x, bn, dtype):
    return (x.is_cuda and dtype in _DT16 and type(bn) is torch.nn.BatchNorm2d and bn.training and bn.affine
            and bn.track_running_stats and bn.momentum is not None and x.shape[1] % 8 == 0
            and x.shape[0] * x.shape[2] * x.shape[3] > 1)


class _BNActTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xh, weight, bias, bn, relu):
        # xh: (B, H, W, C) contiguous, 16-bit
        B, H, W, C = xh.shape
        T = B * H * W
        y = torch.empty_like(xh)
        sums = torch.empty((2, C), dtype=torch.float32, device=xh.device)
        lib = _lib.load_library()
        with on_device(xh.device):
            rc = lib.rfn_bn_train_fwd(ptr(xh), ptr(weight), ptr(bias), ptr(y), ptr(sums), ptr(bn.running_mean),
                                      ptr(bn.running_var), T, C, float(bn.eps), float(bn.momentum), 1 if relu else 0,
                                      _DT16[xh.dtype], current_stream(xh.device))
        _lib.check(rc, "bn_train_fwd")
        bn.num_batches_tracked.add_(1)
        if any(ctx.needs_input_grad[:3]):
            ctx.save_for_backward(xh, sums, weight, bias)
            ctx.eps, ctx.relu = float(bn.eps), relu
        return y

    @staticmethod
    def backward(ctx, gy):
        xh, sums, weight, bias = ctx.saved_tensors
        B, H, W, C = xh.shape
        if gy.dtype != xh.dtype:
            gy = gy.to(xh.dtype)
        if not gy.is_contiguous():
            gy = gy.contiguous()
        gx = torch.empty_like(xh)
        bsums = torch.empty((2, C), dtype=torch.float32, device=xh.device)
        lib = _lib.load_library()
        with on_device(xh.device):
            rc = lib.rfn_bn_train_bwd(ptr(xh), ptr(gy), ptr(sums), ptr(weight), ptr(bias), ptr(gx), ptr(bsums), B * H * W, C,
                                      ctx.eps, 1 if ctx.relu else 0, _DT16[xh.dtype], current_stream(xh.device))
        _lib.check(rc, "bn_train_bwd")
        gw = gb = None
        if ctx.needs_input_grad[1]:
            sink = grad_sink(weight)
            if sink is not None:
                sink.add_(bsums[1])
            else:
                gw = bsums[1].to(weight.dtype)
        if ctx.needs_input_grad[2]:
            sink = grad_sink(bias)
            if sink is not None:
                sink.add_(bsums[0])
            else:
                gb = bsums[0].to(bias.dtype)
        return (gx if ctx.needs_input_grad[0] else None), gw, gb, None, None


def bn_act_train(x, bn, relu, dtype):
    """relu?(bn(x)) with batch statistics; x NCHW-shaped (any memory format, converted to channels-last 16-bit if it is
    not already); returns an NCHW-shaped channels-last tensor."""
    xh = x.permute(0, 2, 3, 1)
    if xh.dtype != dtype:
        xh = xh.to(dtype)
    if not xh.is_contiguous():
        xh = xh.contiguous()
    w = bn.weight if bn.weight.dtype == torch.float32 else bn.weight.float()
    b = bn.bias if bn.bias.dtype == torch.float32 else bn.bias.float()
    return _BNActTrain.apply(xh, w, b, bn, relu).permute(0, 3, 1, 2)
