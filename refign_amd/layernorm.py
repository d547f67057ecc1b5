"""LayerNorm on the HIP kernel of csrc/layernorm.hip (fp32 statistics, fp32/bf16 activations, fwd + bwd).

`LayerNorm` is a drop-in nn.LayerNorm subclass (same parameters / state_dict keys).  On HIP tensors with a 1-D
normalized_shape <= 1024 it runs the hand-written kernel; inside a bf16 autocast region its OUTPUT is bf16 (the library
path returns fp32 there and pays a separate cast pass before every following linear), which also makes the MiT residual
stream bf16.  On CPU tensors (unit tests of the module trees only) it is torch's layer_norm.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._tensor import current_stream, on_device, ptr, workspace
from .params import as_dtype, grad_sink

_DT = {torch.float32: 0, torch.bfloat16: 1}
_LN_WS_ROWS = 256          # kLnMaxBlocks in csrc/layernorm.hip (checked against the ABI in the GPU tests)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        C = x.shape[-1]
        x2 = x.contiguous().view(-1, C)
        rows = x2.shape[0]
        w32 = as_dtype(weight, torch.float32).detach().contiguous()
        b32 = as_dtype(bias, torch.float32).detach().contiguous()
        y = torch.empty((rows, C), dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        lib = _lib.load_library()
        with on_device(x.device):
            rc = lib.rfn_layernorm_fwd(ptr(x2), ptr(w32), ptr(b32), ptr(y), ptr(mean), ptr(rstd), rows, C, float(eps),
                                       _DT[x2.dtype], _DT[out_dtype], current_stream(x.device))
        _lib.check(rc, "layernorm_fwd")
        ctx.save_for_backward(x2, w32, mean, rstd)
        ctx.shape, ctx.wdtype = x.shape, weight.dtype
        ctx.weight, ctx.bias = weight, bias
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        return _ln_backward(ctx, gy, None)


def _ln_backward(ctx, gy, add, gy_b=None):
        x2, w32, mean, rstd = ctx.saved_tensors
        rows, C = x2.shape
        if gy is None:
            gy, gy_b = gy_b, None
        if gy is None:                                     # only the pass-through output was used
            return (add, None, None, None, None)
        if gy.dtype not in _DT:
            gy = gy.float()
        gy2 = gy.contiguous().view(rows, C)
        if gy_b is not None:                               # a second consumer of the LayerNorm output (layer_norm_pass2)
            if C % 8 == 0:
                gy_b = gy_b.to(gy2.dtype).contiguous().view(rows, C)
            else:
                gy2, gy_b = gy2 + gy_b.to(gy2.dtype).contiguous().view(rows, C), None
        dx = torch.empty_like(x2)
        if add is not None:
            add = add.to(x2.dtype).contiguous().view(rows, C)
        # parameter gradients: straight into the flat gradient buffer when the trainer provides one
        sg, sb = grad_sink(ctx.weight), grad_sink(ctx.bias)
        direct = sg is not None and sb is not None
        dg = sg if direct else torch.empty(C, dtype=torch.float32, device=x2.device)
        db = sb if direct else torch.empty(C, dtype=torch.float32, device=x2.device)
        lib = _lib.load_library()
        ws = workspace(_LN_WS_ROWS * 2 * C * 4, x2.device)     # == rfn_layernorm_bwd_workspace_bytes(C)
        with on_device(x2.device):
            if gy_b is not None:
                rc = lib.rfn_layernorm_bwd_add2(ptr(x2), ptr(gy2), ptr(gy_b), ptr(add) if add is not None else None, ptr(w32),
                                                ptr(mean), ptr(rstd), ptr(dx), ptr(dg), ptr(db), ptr(ws), rows, C, _DT[x2.dtype],
                                                _DT[gy2.dtype], 1 if direct else 0, current_stream(x2.device))
                add = None
            elif add is not None and C % 8 == 0:
                rc = lib.rfn_layernorm_bwd_add(ptr(x2), ptr(gy2), ptr(add), ptr(w32), ptr(mean), ptr(rstd), ptr(dx), ptr(dg),
                                               ptr(db), ptr(ws), rows, C, _DT[x2.dtype], _DT[gy2.dtype], 1 if direct else 0,
                                               current_stream(x2.device))
                add = None
            else:
                rc = lib.rfn_layernorm_bwd(ptr(x2), ptr(gy2), ptr(w32), ptr(mean), ptr(rstd), ptr(dx), ptr(dg), ptr(db),
                                           ptr(ws), rows, C, _DT[x2.dtype], _DT[gy2.dtype], 1 if direct else 0,
                                           current_stream(x2.device))
        _lib.check(rc, "layernorm_bwd")
        if add is not None:
            dx = dx + add
        if direct:
            return dx.view(ctx.shape), None, None, None, None
        return dx.view(ctx.shape), dg.to(ctx.wdtype), db.to(ctx.wdtype), None, None


class _LayerNormPassFn(torch.autograd.Function):
    """(LayerNorm(x), x) -- the second output is x itself, to be used as the residual operand of the branch that follows
    (mix_transformer.py:203-207: `x + drop_path(f(norm(x)))`).  Both gradients then arrive HERE and are summed inside the
    LayerNorm-backward kernel (rfn_layernorm_bwd_add) instead of by an element-wise kernel of the autograd engine."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        y = _LayerNormFn.forward(ctx, x, weight, bias, eps, out_dtype)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gx_res):
        return _ln_backward(ctx, gy, gx_res)


class _LayerNormPass2Fn(torch.autograd.Function):
    """(LayerNorm(x), LayerNorm(x) again, x) -- the LayerNorm output for TWO consumers (in a MiT attention block norm1(x) feeds the
    q projection and the spatial-reduction convolution, mix_transformer.py:142-150) plus the pass-through of _LayerNormPassFn:
    all three gradients arrive here and are summed inside the LayerNorm-backward kernel (rfn_layernorm_bwd_add2)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        y = _LayerNormFn.forward(ctx, x, weight, bias, eps, out_dtype)
        return y, y.view_as(y), x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gy_b, gx_res):
        return _ln_backward(ctx, gy, gx_res, gy_b)


def layer_norm_pass2(x, weight, bias, eps=1e-5):
    """-> (LayerNorm(x), LayerNorm(x), x): two handles on the output for two consumers, and the residual pass-through."""
    out_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
    if out_dtype not in _DT:
        out_dtype = x.dtype
    return _LayerNormPass2Fn.apply(x, weight, bias, eps, out_dtype)


def layer_norm_pass(x, weight, bias, eps=1e-5):
    """-> (LayerNorm(x), x) with the residual gradient folded into the LayerNorm backward (HIP tensors, C % 8 == 0)."""
    out_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
    if out_dtype not in _DT:
        out_dtype = x.dtype
    return _LayerNormPassFn.apply(x, weight, bias, eps, out_dtype)


def layer_norm(x, weight, bias, eps=1e-5, out_dtype=None):
    if x.dtype not in _DT:
        x = x.float()
    if out_dtype is None:
        out_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        if out_dtype not in _DT:
            out_dtype = x.dtype
    return _LayerNormFn.apply(x, weight, bias, eps, out_dtype)


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        if x.is_cuda and len(self.normalized_shape) == 1 and self.normalized_shape[0] <= 1024 \
                and self.weight is not None and self.bias is not None:
            return layer_norm(x, self.weight, self.bias, self.eps)
        return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
