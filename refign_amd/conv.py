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
