"""Depthwise 3x3 convolution on channels-last / token layout, forward and backward in HIP (csrc/dwconv.hip).

`dwconv3x3_tokens(x, weight, bias, H, W)` is the DWConv of the Mix-FFN (mix_transformer.py:556-568) WITHOUT the two
NCHW transposes of the reference: x is (B, N=H*W, C) and stays that way.  `dwconv3x3_nhwc(x, weight, bias, dilation)`
is the same on (B, H, W, C) maps with dilation (DAFormer ASPP branches, daformer.py:46-62).  `weight` is the reference
parameter itself, shape (C, 1, 3, 3); activations float32 or bfloat16, accumulation fp32, weight grads fp32.
"""
import os

import torch

from . import _lib
from ._tensor import current_stream, on_device, ptr, require_device_tensor, workspace
from .params import as_dtype, derived, grad_sink

_DT = {torch.float32: 0, torch.bfloat16: 1}
_DW_WS_STRIPES = 128       # kMaxStripes in csrc/dwconv.hip (checked against the ABI in the GPU tests)


def _fwd(x, w_tap, bias, dilation, flip, stats=None):
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    lib = _lib.load_library()
    if stats is not None:                                   # + the BatchNorm statistics of the result (csrc/dwconv.hip STATS)
        with on_device(x.device):
            rc = lib.rfn_dwconv3x3_nhwc_fwd_stats(ptr(x), ptr(w_tap), ptr(bias), ptr(y), ptr(stats), B, H, W, C, dilation,
                                                  _DT[x.dtype], current_stream(x.device))
        _lib.check(rc, "dwconv3x3_nhwc_fwd_stats")
        return y
    with on_device(x.device):
        rc = lib.rfn_dwconv3x3_nhwc_fwd(ptr(x), ptr(w_tap), ptr(bias), ptr(y), B, H, W, C, dilation, _DT[x.dtype],
                                        1 if flip else 0, current_stream(x.device))
    _lib.check(rc, "dwconv3x3_nhwc_fwd")
    return y


class _DWConv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, dilation, stats=None):
        if x.dtype not in _DT:
            x = x.float()
        x = require_device_tensor(x.contiguous(), "x")
        C = x.shape[-1]
        # (9, C) tap-major fp32 copy of the parameter, re-made only when the parameter changed
        w_tap = derived(weight, "tap_major_f32", lambda t: t.float().reshape(C, 9).t().contiguous(),
                        lambda t: t.reshape(C, 9).t())
        b32 = None if bias is None else as_dtype(bias, torch.float32).detach().contiguous()
        ctx.save_for_backward(x, w_tap)
        ctx.dilation, ctx.has_bias = dilation, bias is not None
        ctx.wshape, ctx.wdtype = weight.shape, weight.dtype
        ctx.weight, ctx.bias = weight, bias
        return _fwd(x, w_tap, b32, dilation, False, stats)

    @staticmethod
    def backward(ctx, gy):
        x, w_tap = ctx.saved_tensors
        return _dwconv_backward(ctx, x, w_tap, gy.to(x.dtype).contiguous()) + (None,)


class _DWConv3x3Gelu(torch.autograd.Function):
    """gelu(dwconv3x3(x)) in one kernel; the backward is GELU' (on the saved pre-activation) followed by the backward
    of the plain convolution."""

    @staticmethod
    def forward(ctx, x, weight, bias, with_z=False):
        if x.dtype not in _DT:
            x = x.float()
        x = require_device_tensor(x.contiguous(), "x")
        B, H, W, C = x.shape
        w_tap = derived(weight, "tap_major_f32", lambda t: t.float().reshape(C, 9).t().contiguous(),
                        lambda t: t.reshape(C, 9).t())
        b32 = None if bias is None else as_dtype(bias, torch.float32).detach().contiguous()
        need = any(ctx.needs_input_grad)
        with_z = with_z and need
        ctx.set_materialize_grads(False)          # the non-differentiable output's "gradient" must not become a zero tensor
        z = torch.empty_like(x) if need else None
        a = torch.empty_like(x)
        lib = _lib.load_library()
        with on_device(x.device):
            rc = lib.rfn_dwconv3x3_gelu_nhwc_fwd(ptr(x), ptr(w_tap), ptr(b32), ptr(z), ptr(a), B, H, W, C, _DT[x.dtype],
                                                 current_stream(x.device))
        _lib.check(rc, "dwconv3x3_gelu_nhwc_fwd")
        if need:
            ctx.save_for_backward(x, w_tap, z)
            ctx.has_bias, ctx.wshape, ctx.wdtype = bias is not None, weight.shape, weight.dtype
            ctx.weight, ctx.bias, ctx.dilation = weight, bias, 1
        if with_z:
            # (a, z): the activation as a NON-differentiable tensor + the pre-activation that carries the gradient -- the
            # consumer (linear._LinearFn with z=) returns d/dz directly, gelu' applied in its input-gradient GEMM's epilogue
            ctx.mark_non_differentiable(a)
            ctx.with_z = True
            return a, z
        ctx.with_z = False
        return a

    @staticmethod
    def backward(ctx, *grads):
        x, w_tap, z = ctx.saved_tensors
        if ctx.with_z:
            if grads[1] is None:
                return None, None, None, None
            gz = grads[1].to(z.dtype).contiguous()
        else:
            gz = torch.ops.aten.gelu_backward(grads[0].to(z.dtype).contiguous(), z)
        ctx.saved = (x, w_tap)
        return _dwconv_backward(ctx, x, w_tap, gz)[:3] + (None,)


def _dwconv_backward(ctx, x, w_tap, gy):
    B, H, W, C = x.shape
    gx = gw = gb = None
    if ctx.needs_input_grad[0]:
        gx = _fwd(gy, w_tap, None, ctx.dilation, True)
    if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
        sw, sb = grad_sink(ctx.weight), grad_sink(ctx.bias) if ctx.has_bias else None
        direct = sw is not None and (sb is not None or not ctx.has_bias)
        if direct:
            dw, db = sw, sb
        else:
            dw = torch.empty((9, C), dtype=torch.float32, device=x.device)
            db = torch.empty((C,), dtype=torch.float32, device=x.device) if ctx.has_bias else None
        lib = _lib.load_library()

        def run():
            ws = workspace(_DW_WS_STRIPES * 10 * C * 4, x.device)        # per stream: looked up on the stream it runs on
            with on_device(x.device):
                rc = lib.rfn_dwconv3x3_nhwc_bwd_weight(ptr(x), ptr(gy), ptr(dw), ptr(db), ptr(ws), B, H, W, C,
                                                       ctx.dilation, _DT[x.dtype], 3 if direct else 0,
                                                       current_stream(x.device))
            _lib.check(rc, "dwconv3x3_nhwc_bwd_weight")

        run()
        if not direct:
            gw = dw.t().reshape(ctx.wshape).to(ctx.wdtype)
            gb = db
    return gx, gw, gb, None


def dwconv3x3_gelu_tokens(x, weight, bias, H, W, with_z=False):
    """gelu(DWConv(x)) on tokens (B, N=H*W, C) -> (B, N, C): mix_transformer.py:99-101 in one pass.  with_z (under autograd):
    -> (a, z), a non-differentiable, z the pre-activation carrying the gradient (see _DWConv3x3Gelu); (a, None) otherwise."""
    B, N, C = x.shape
    out = _DWConv3x3Gelu.apply(x.reshape(B, H, W, C), weight, bias, with_z)
    if isinstance(out, tuple):
        return out[0].reshape(B, N, C), out[1].reshape(B, N, C)
    return (out.reshape(B, N, C), None) if with_z else out.reshape(B, N, C)


FUSED_FFN = os.environ.get("RFN_FUSED_FFN", "1") != "0"      # (tests flip the attribute; the variable is for A/B runs of bench.py)


@torch.no_grad()
def ffn_fc1_dw_gelu(x, fc1, dw, H, W):
    """gelu(dw(fc1(x))) of a Mix-FFN (mix_transformer.py:99-101) on gradient-free bf16 tokens (views, H*W, C) in ONE kernel
    (csrc/mixffn.hip): the 4C-wide pre-activation never reaches HBM.  `fc1`: the Linear, `dw`: the depthwise nn.Conv2d.  None
    outside the kernel's domain (the caller runs fc1, then dwconv3x3_gelu_tokens)."""
    B, N, C = x.shape
    HID = fc1.weight.shape[0]
    if not (FUSED_FFN and x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and N == H * W and C % 64 == 0
            and HID % 128 == 0 and fc1.bias is not None and dw.bias is not None and dw.weight.shape == (HID, 1, 3, 3)
            and dw.padding == (1, 1) and dw.stride == (1, 1) and dw.dilation == (1, 1)):
        return None
    w1, b1 = as_dtype(fc1.weight, torch.bfloat16), as_dtype(fc1.bias, torch.bfloat16)
    w_tap = derived(dw.weight, "tap_major_f32", lambda t: t.float().reshape(HID, 9).t().contiguous(), lambda t: t.reshape(HID, 9).t())
    bdw = as_dtype(dw.bias, torch.float32).detach().contiguous()
    a = torch.empty((B, N, HID), dtype=torch.bfloat16, device=x.device)
    with on_device(x.device):
        rc = _lib.load_library().rfn_ffn_fc1_dw_gelu_bf16(ptr(x), ptr(w1), ptr(b1), ptr(w_tap), ptr(bdw), ptr(a), B, H, W, C, HID,
                                                          current_stream(x.device))
    _lib.check(rc, "ffn_fc1_dw_gelu")
    return a


def dwconv3x3_nhwc(x, weight, bias=None, dilation=1, stats=None):
    """x: (B,H,W,C) fp32/bf16; weight: (C,1,3,3); bias: (C) or None; same-size output (padding = dilation).
    `stats` (bf16 x, C % 8 == 0): a float64 tensor of 2 C + 1 elements that receives the BatchNorm statistics of the result
    (sum, sum of squares, rows: the buffer of bn._stats_fwd)."""
    if x.dim() != 4 or weight.shape[0] != x.shape[-1]:
        raise RuntimeError("dwconv3x3_nhwc: x must be (B,H,W,C) and weight (C,1,3,3)")
    if stats is not None and not (x.dtype == torch.bfloat16 and x.shape[-1] % 8 == 0 and stats.dtype == torch.float64
                                  and stats.is_contiguous() and stats.numel() == 2 * x.shape[-1] + 1):
        raise RuntimeError("dwconv3x3_nhwc(stats=...): bf16 input with C % 8 == 0 and a float64 buffer of 2 C + 1 elements")
    return _DWConv3x3.apply(x, weight, bias, int(dilation), stats)


@torch.no_grad()
def dwconv3x3_bn_act_nhwc(x, weight, bias, dilation, bn, relu):
    """act(bn(dwconv3x3(x))) with BATCH statistics, gradient-free (the EMA teacher's ASPP branches run their BatchNorms in
    training mode; daformer.py:10-62): two passes over x -- statistics of the convolution result without storing
    it, then convolution + normalisation + ReLU -- instead of convolution, statistics pass and BatchNorm pass over the
    result.  x: (B, H, W, C) bf16 contiguous; bn: the (Sync)BatchNorm2d module (running buffers updated as in training)."""
    from . import bn as bnk
    B, H, W, C = x.shape
    w_tap = derived(weight, "tap_major_f32", lambda t: t.float().reshape(C, 9).t().contiguous(),
                    lambda t: t.reshape(C, 9).t())
    b32 = None if bias is None else as_dtype(bias, torch.float32).detach().contiguous()
    g = None if bn.weight is None else as_dtype(bn.weight, torch.float32).detach().contiguous()
    be = None if bn.bias is None else as_dtype(bn.bias, torch.float32).detach().contiguous()
    sums = torch.empty(2 * C + 1, dtype=torch.float64, device=x.device)
    y = torch.empty_like(x)
    lib = _lib.load_library()
    with on_device(x.device):
        rc = lib.rfn_dwconv3x3_nhwc_stats(ptr(x), ptr(w_tap), ptr(b32), ptr(sums), B, H, W, C, int(dilation), _DT[x.dtype],
                                          current_stream(x.device))
    _lib.check(rc, "dwconv3x3_nhwc_stats")
    group = bnk.sync_group(bn)
    if group is not None:
        bnk._all_reduce(sums, group, bnk._exchange_comm(bn))
    with on_device(x.device):
        rc = lib.rfn_dwconv3x3_bn_act_nhwc_fwd(ptr(x), ptr(w_tap), ptr(b32), ptr(g), ptr(be), ptr(sums), ptr(bn.running_mean),
                                               ptr(bn.running_var), ptr(y), B, H, W, C, int(dilation), float(bn.eps),
                                               float(bn.momentum), 1 if relu else 0, _DT[x.dtype], current_stream(x.device))
    _lib.check(rc, "dwconv3x3_bn_act_nhwc_fwd")
    bn.num_batches_tracked.add_(1)
    return y


def tri_usable(x, convs, bns):
    """the three (conv, bn) pairs are depthwise 3x3 with dilations g, 2 g, 3 g and padding = dilation on a shape the one-pass
    kernel takes (csrc/dwconv.hip dwconv3x3_tri_kernel)"""
    if len(convs) != 3 or not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous()):
        return False
    B, H, W, C = x.shape
    d = [c.dilation[0] for c in convs]
    ok = all(c.groups == c.in_channels == c.out_channels == C and c.kernel_size == (3, 3) and c.stride == (1, 1)
             and c.padding == c.dilation and c.dilation[0] == c.dilation[1] for c in convs)
    ok = ok and d[1] == 2 * d[0] and d[2] == 3 * d[0] and all(b.momentum is not None for b in bns)
    same_bias = all((c.bias is None) == (convs[0].bias is None) for c in convs)
    return bool(ok and same_bias and _lib.load_library().rfn_dwconv3x3_tri_usable(B, H, W, C, d[0]))


@torch.no_grad()
def dwconv3x3_bn_act_nhwc_tri(x, convs, bns, relu):
    """[act(bn_k(dwconv3x3_k(x))) for k in 0..2] with BATCH statistics, gradient-free, for three depthwise branches of dilations
    g, 2 g, 3 g of one input (the EMA teacher's ASPP): TWO passes over x in all -- statistics of the three results, then the
    three convolutions + normalisation + ReLU -- where dwconv3x3_bn_act_nhwc makes six."""
    import ctypes
    from . import bn as bnk
    B, H, W, C = x.shape
    g = convs[0].dilation[0]
    w3 = torch.stack([derived(c.weight, "tap_major_f32", lambda t: t.float().reshape(C, 9).t().contiguous(),
                              lambda t: t.reshape(C, 9).t()) for c in convs]).contiguous()
    b3 = None if convs[0].bias is None else torch.stack([as_dtype(c.bias, torch.float32).detach() for c in convs]).contiguous()
    sums = torch.empty((3, 2 * C + 1), dtype=torch.float64, device=x.device)
    lib = _lib.load_library()
    with on_device(x.device):
        rc = lib.rfn_dwconv3x3_tri_stats(ptr(x), ptr(w3), ptr(b3), ptr(sums), B, H, W, C, int(g), current_stream(x.device))
    _lib.check(rc, "dwconv3x3_tri_stats")
    for k, bn in enumerate(bns):
        group = bnk.sync_group(bn)
        if group is not None:
            bnk._all_reduce(sums[k], group, bnk._exchange_comm(bn))
    ys = [torch.empty_like(x) for _ in range(3)]
    keep = []                                              # fp32 views of the affine parameters stay alive until the launch

    def f32(t):
        if t is None:
            return None
        keep.append(as_dtype(t, torch.float32).detach().contiguous())
        return keep[-1]

    arr = lambda ts: (ctypes.c_void_p * 3)(*[ptr(t) for t in ts])  # noqa: E731
    ga, be = arr([f32(b.weight) for b in bns]), arr([f32(b.bias) for b in bns])
    rm, rv = arr([b.running_mean for b in bns]), arr([b.running_var for b in bns])
    yp = arr(ys)
    eps = (ctypes.c_float * 3)(*[float(b.eps) for b in bns])
    mom = (ctypes.c_float * 3)(*[float(b.momentum) for b in bns])
    with on_device(x.device):
        rc = lib.rfn_dwconv3x3_tri_bn_act_fwd(ptr(x), ptr(w3), ptr(b3), ga, be, ptr(sums), rm, rv, yp, B, H, W, C, int(g), eps, mom,
                                              1 if relu else 0, current_stream(x.device))
    _lib.check(rc, "dwconv3x3_tri_bn_act_fwd")
    for b in bns:
        b.num_batches_tracked.add_(1)
    return ys


def dwconv3x3_tokens(x, weight, bias, H, W):
    """x: (B, N=H*W, C) tokens -> (B, N, C)."""
    B, N, C = x.shape
    return dwconv3x3_nhwc(x.reshape(B, H, W, C), weight, bias, 1).reshape(B, N, C)
