"""nn.Conv2d for the MiT patch-embedding / spatial-reduction convolutions, with the parameter plumbing of params.py.

The convolution itself (forward, data gradient, weight gradient) is the ROCm library's; what changes is how the
parameters enter and leave it: the weight is used through its cached bf16 copy instead of being re-cast by autocast at
every call (80 spatial-reduction convs x 7 uses per step), and the weight / bias gradients are added straight into
the fp32 views of the flat gradient buffer (no bf16 -> fp32 cast kernel + AccumulateGrad add per use).
Same parameters / state_dict keys as nn.Conv2d; CPU tensors take the stock path.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._tensor import current_stream, on_device, ptr
from .params import _identity, as_dtype, compute_dtype, derived, grad_sink


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias, w_c, b_c, stride, padding, dilation, groups):
        ctx.save_for_backward(x, w_c)
        ctx.weight, ctx.bias = weight, bias
        ctx.conf = (stride, padding, dilation, groups)
        return F.conv2d(x, w_c, b_c, stride, padding, dilation, groups)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, w_c = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.conf
        has_bias = ctx.bias is not None
        mask = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]]
        gx, gw, gb = torch.ops.aten.convolution_backward(
            gy.to(w_c.dtype), x, w_c, [w_c.shape[0]] if has_bias else None, list(stride), list(padding),
            list(dilation), False, [0, 0], groups, mask)
        if gw is not None:
            sink = grad_sink(ctx.weight)
            if sink is not None:
                sink.add_(gw)
                gw = None
            else:
                gw = gw.to(ctx.weight.dtype)
        if gb is not None:
            sink = grad_sink(ctx.bias)
            if sink is not None:
                sink.add_(gb)
                gb = None
            else:
                gb = gb.to(ctx.bias.dtype)
        return gx, gw, gb, None, None, None, None, None, None


class Conv2d(nn.Conv2d):
    def forward(self, x):
        if not x.is_cuda or self.padding_mode != 'zeros' or isinstance(self.padding, str):
            return super().forward(x)
        cd = compute_dtype(x)
        # the MiT token tensors reach the convolutions as channels-last views, and the library then wants the filter in
        # channels-last too: keep the cached copy in that layout instead of converting it at every call
        w_c = derived(self.weight, (cd, "channels_last"),
                      lambda t: t.to(cd).contiguous(memory_format=torch.channels_last), _identity)
        b_c = as_dtype(self.bias, cd)
        if x.dtype != cd:
            x = x.to(cd)
        if torch.is_grad_enabled() and self.weight.requires_grad:
            return _Conv2dFn.apply(x, self.weight, self.bias, w_c, b_c, self.stride, self.padding, self.dilation,
                                   self.groups)
        return F.conv2d(x, w_c, b_c, self.stride, self.padding, self.dilation, self.groups)


# ---------------------------------------------------------------------------------------------------------------------
# Spatial-reduction convolution of the MiT attention (kernel = stride = sr_ratio, no padding: mix_transformer.py:128-134)
# as what it is -- a Linear over non-overlapping r x r patches -- on the token layout, without the NCHW round trip.
# Forward: one gather copy (tokens -> patches) + one GEMM with bias, output directly as (B, N', C) tokens for the
# LayerNorm that follows.  Backward: the Linear's three GEMM-shaped ops (split-T weight gradient, parameter gradients
# accumulated into the flat buffer) + one scatter copy; the library's convolution backward for this shape is five
# launches of its own plus per-call zero-fill / cast tensor ops (~85 us of GPU time per layer per pass, 240 per step).
# ---------------------------------------------------------------------------------------------------------------------
_DT = {torch.float32: 0, torch.bfloat16: 1}


def _patchify(src, dst, B, H, W, C, r, inverse):
    lib = _lib.load_library()
    with on_device(src.device):
        rc = lib.rfn_patchify_tokens(ptr(src), ptr(dst), B, H, W, C, r, _DT[src.dtype], 1 if inverse else 0,
                                     current_stream(src.device))
    _lib.check(rc, "patchify_tokens")


def _to_patches(x, H, W, r):
    B, N, C = x.shape
    Hr, Wr = H // r, W // r
    if x.is_cuda and x.dtype in _DT and C % 8 == 0 and x.is_contiguous():
        out = torch.empty((B * Hr * Wr, r * r * C), dtype=x.dtype, device=x.device)
        _patchify(x, out, B, H, W, C, r, False)                      # one vectorised gather (csrc/upcat.hip)
        return out, Hr, Wr
    v = x.view(B, H, W, C)
    if Hr * r != H or Wr * r != W:
        v = v[:, :Hr * r, :Wr * r]                   # the strided conv drops the ragged border
    return v.reshape(B, Hr, r, Wr, r, C).permute(0, 1, 3, 2, 4, 5).reshape(B * Hr * Wr, r * r * C), Hr, Wr


def _from_patches(gp, B, H, W, C, r, Hr, Wr):
    if gp.is_cuda and gp.dtype in _DT and C % 8 == 0 and gp.is_contiguous():
        ragged = Hr * r != H or Wr * r != W
        out = (torch.zeros if ragged else torch.empty)((B, H * W, C), dtype=gp.dtype, device=gp.device)
        _patchify(gp, out, B, H, W, C, r, True)
        return out
    g = gp.view(B, Hr, Wr, r, r, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hr * r, Wr * r, C)
    if Hr * r != H or Wr * r != W:
        g = F.pad(g, (0, 0, 0, W - Wr * r, 0, H - Hr * r))
    return g.reshape(B, H * W, C)


class _PatchLinearFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias, w2, b_c, H, W, r):
        patches, Hr, Wr = _to_patches(x, H, W, r)
        ctx.save_for_backward(patches, w2)
        ctx.weight, ctx.bias = weight, bias
        ctx.geom = (x.shape, H, W, r, Hr, Wr)
        return F.linear(patches, w2, b_c).view(x.shape[0], Hr * Wr, w2.shape[0])

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        from .linear import _split
        from .params import sum_rows
        patches, w2 = ctx.saved_tensors
        (B, N, C), H, W, r, Hr, Wr = ctx.geom
        Co, K = w2.shape
        g2 = gy.reshape(-1, Co)
        if g2.dtype != w2.dtype:
            g2 = g2.to(w2.dtype)
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _from_patches(torch.mm(g2, w2), B, H, W, C, r, Hr, Wr)
        T = g2.shape[0]
        S = _split(T)
        if S > 1:
            part = torch.bmm(g2.view(S, T // S, Co).transpose(1, 2), patches.view(S, T // S, K)).view(S, Co * K)
        else:
            part = g2.t().mm(patches).view(1, Co * K)
        sw, sb = grad_sink(ctx.weight), grad_sink(ctx.bias)
        # the parameter is (Co, C, r, r), the GEMM's weight is its (Co, r, r, C) permutation
        gw2 = sum_rows(part).view(Co, r, r, C)
        if sw is not None:
            sw.permute(0, 2, 3, 1).add_(gw2)
        else:
            gw = gw2.permute(0, 3, 1, 2).to(ctx.weight.dtype)
        if ctx.bias is not None:
            if sb is not None:
                sum_rows(g2, out=sb, accumulate=True)
            else:
                gb = sum_rows(g2).to(ctx.bias.dtype)
        return gx, gw, gb, None, None, None, None, None


def patch_conv_tokens(x, H, W, conv):
    """`conv` (kernel == stride, no padding, groups 1) applied to the (B, H*W, C) token map `x`; returns the
    (B, (H//r)*(W//r), C_out) token map -- what `conv(x.transpose(1, 2).reshape(B, C, H, W)).flatten(2).transpose(1, 2)`
    gives.  None when the layer / tensor is outside this path's domain."""
    r = conv.kernel_size[0]
    if not (x.is_cuda and conv.kernel_size == (r, r) and conv.stride == (r, r) and conv.padding == (0, 0)
            and conv.dilation == (1, 1) and conv.groups == 1 and H >= r and W >= r and x.is_contiguous()):
        return None
    cd = compute_dtype(x)
    Co, C = conv.weight.shape[:2]
    w2 = derived(conv.weight, (cd, "patch_linear"), lambda t: t.to(cd).permute(0, 2, 3, 1).contiguous(),
                 lambda t: t.permute(0, 2, 3, 1)).view(Co, r * r * C)
    b_c = as_dtype(conv.bias, cd)
    if x.dtype != cd:
        x = x.to(cd)
    if torch.is_grad_enabled() and conv.weight.requires_grad:
        return _PatchLinearFn.apply(x, conv.weight, conv.bias, w2, b_c, H, W, r)
    patches, Hr, Wr = _to_patches(x, H, W, r)
    return F.linear(patches, w2, b_c).view(x.shape[0], Hr * Wr, Co)
