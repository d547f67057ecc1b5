"""nn.Linear with the weight-gradient path laid out for MI355X and the flat gradient buffer.

The weight gradient of a token-wise Linear is dW[N,K] = dY[T,N]^T . X[T,K] with T = tokens (8 160 ... 259 200 on the
Refign step) and N, K <= 2048: a "small output, very long reduction" GEMM.  The library picks a 64x64 macro-tile without
split-K for it, i.e. 25-400 workgroups each walking the whole T: measured 24 TF/s on [320 x 8160] x [8160 x 320]
(240 of those per step, 58 ms of wgrad GEMMs in total, profiles/r01_step_shapes_bf16.txt).  Splitting T into
S independent slabs turns it into a batched GEMM with S x more workgroups; the (S, N, K) partials are reduced by
csrc/reduce.hip, which ADDS the result straight into the parameter's .grad view of the flat gradient buffer; the bias
gradient (column sum of dY) goes through the same kernel.  Weights are used through their cached bf16 copies
(params.derived) instead of being re-cast at every call.  Forward and the input gradient run on the hand-written MFMA GEMMs
(mfma.gemm_nt: first- and second-generation kernels, bias / activation / residual in the epilogue; fp32 tensors: split32.linear);
CPU tensors and shapes outside the kernels' domain go to ATen (F.linear), recorded on a GPU by mfma.note_library.
Same parameters / state_dict keys as nn.Linear.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import mfma
from .params import as_dtype, compute_dtype, grad_sink, linear_param_grads, sum_rows, transposed


_FUSED_GRADS = True
# residual + stochastic depth in the proj / fc2 GEMMs under autograd (the gradient-free passes always fuse).  Round 2
# measured it neutral (239.0 / 241.5 vs 237.5 / 240.1 ms per step); with the student passes replayed from graphs and the
# library kernels gone, the 410 element-wise launches it removes are worth 1.0-1.2 ms per step (181.4 / 181.5 vs 182.4 /
# 182.7 ms, round 3): ON by default, RFN_FUSED_RESIDUAL=0 switches it off.
_FUSED_RESIDUAL = os.environ.get("RFN_FUSED_RESIDUAL", "1") != "0"


def _split(T):
    for s in (64, 32, 16, 8, 4, 2):
        if T % s == 0 and T // s >= 384:
            return s
    return 1


class _LinearFn(torch.autograd.Function):
    """x, w_c, b_c are in the compute dtype; weight / bias are the fp32 parameters (gradient routing only)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias, w_c, b_c, res=None, rowscale=None, z=None):
        """With `res` (same shape as the output, compute dtype) and optionally `rowscale` ((B,) fp32, one value per
        leading-dimension sample): y = res + rowscale[b] * (x W^T + b) in the GEMM epilogue -- the stochastic-depth
        residual of mix_transformer.py:203-207 under autograd.  Backward: the residual's gradient is the incoming one;
        the branch's gradient diag(rowscale) g is never materialised: the scale rides in the epilogue of the input-
        gradient GEMM and in the operand staging of the weight-gradient kernel."""
        # z (optional): x = gelu(z) was produced by the depthwise + GELU kernel of the Mix-FFN (mix_transformer.py:99-102), which
        # hands over x as a NON-differentiable tensor together with its pre-activation; the gradient this function returns
        # is then the one with respect to z, (dy W) * gelu'(z), made in the epilogue of the input-gradient GEMM (act = 4) --
        # no gelu_backward pass over the 4C-wide hidden tensor
        ctx.weight, ctx.bias = weight, bias
        N, K = w_c.shape
        x2 = x.reshape(-1, K)
        ctx.rps = x2.shape[0] // x.shape[0] if rowscale is not None else 0
        ctx.z = z
        if res is not None:
            ctx.save_for_backward(x, w_c, rowscale)
            y = mfma.gemm_nt(x2, w_c, b_c, res=res.reshape(-1, N), rowscale=rowscale, rows_per_sample=ctx.rps)
            if y is None:
                raise RuntimeError("fused residual Linear: operands outside the MFMA kernel's domain")
            return y.view(x.shape[:-1] + (N,))
        ctx.save_for_backward(x, w_c, None)
        y = mfma.gemm_nt(x2, w_c, b_c)                       # hand-written MFMA kernel (16-bit operands)
        if y is None:
            mfma.note_library("linear.fwd", x2, w_c)
        return F.linear(x, w_c, b_c) if y is None else y.view(x.shape[:-1] + (N,))

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, w_c, rowscale = ctx.saved_tensors
        N, K = w_c.shape
        gx = gw = gb = None
        z = ctx.z
        want_x = ctx.needs_input_grad[0] or (z is not None and ctx.needs_input_grad[7])
        g_res = gy if ctx.needs_input_grad[5] else None

        def route(t):          # (gx, gw, gb, None, None, g_res, None, gz): the input gradient belongs to z when z was given
            return (None,) + t[1:5] + (g_res, None, t[0]) if z is not None else t[:5] + (g_res, None, None)
        if rowscale is not None:
            out = _LinearFn._backward_scaled(ctx, gy, x, w_c, rowscale, want_x)
            if out is not None:
                return route(out)
            gy = gy * rowscale.to(gy.dtype).view((-1,) + (1,) * (gy.dim() - 1))
        if gy.dtype != w_c.dtype:
            gy = gy.to(w_c.dtype)
        g2 = gy.reshape(-1, N)
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        if want_x:
            # dx = dy W = dy (W^T)^T: the NT kernel on the cached transposed 16-bit copy of the weight
            gx = None
            if mfma.ENABLED and w_c.dtype != torch.float32:
                if z is not None:
                    gx = mfma.gemm_nt(g2, transposed(ctx.weight, w_c.dtype), res=z.reshape(-1, K), act=4)
                    if gx is not None:
                        z = None                           # gelu' applied in the epilogue
                else:
                    gx = mfma.gemm_nt(g2, transposed(ctx.weight, w_c.dtype))
            if gx is None:
                gx = mfma.gemm_nt(g2, transposed(ctx.weight, w_c.dtype)) if mfma.ENABLED and w_c.dtype != torch.float32 \
                    else None
            if gx is None:
                mfma.note_library("linear.dgrad", g2, w_c)
            gx = (torch.mm(g2, w_c) if gx is None else gx).view(x.shape)
            if z is not None:
                gx = torch.ops.aten.gelu_backward(gx, z.to(gx.dtype))
            z = ctx.z
        need_w, need_b = ctx.needs_input_grad[1], ctx.bias is not None and ctx.needs_input_grad[2]
        sink_w, sink_b = grad_sink(ctx.weight), grad_sink(ctx.bias)
        part = None
        if need_w and sink_w is not None and (not need_b or sink_b is not None):
            # hand-written split-T MFMA kernel accumulating straight into the flat gradient buffer (weight AND bias).
            # (Forking it to a second stream inside the captured pass was measured and dropped in round 2: 449 vs 244 ms per
            # step -- hipGraph replays cross-stream branches with a synchronisation per edge.)
            x2 = x.reshape(-1, K)
            # (inside the trainer's backward: queued and launched with the block's other weight gradients, mfma.deferred_wgrads)
            if mfma.defer_gemm_tn(g2, x2, sink_w, sink_b if need_b else None) or \
                    mfma.gemm_tn(g2, x2, out=sink_w, bias_out=sink_b if need_b else None) is not None:
                return route((gx, None, None, None, None))
        if need_w:
            x2 = x.reshape(-1, K)
            T = x2.shape[0]
            part = mfma.gemm_tn(g2, x2)                      # fp32 per-slab partials (S, N, K)
            if part is not None:
                part = part.view(part.shape[0], N * K)
            S = _split(T)
            if part is None:
                mfma.note_library("linear.wgrad", g2, x2)
                if S > 1:
                    part = torch.bmm(g2.view(S, T // S, N).transpose(1, 2), x2.view(S, T // S, K)).view(S, N * K)
                else:
                    part = g2.t().mm(x2).view(1, N * K)
        # both gradients straight into the flat gradient buffer in two launches
        if need_w and need_b and sink_w is not None and sink_b is not None and _FUSED_GRADS and \
                part.dtype == g2.dtype and linear_param_grads(g2, part, sink_b, sink_w.view(-1)):
            return route((gx, None, None, None, None))
        if need_w:
            if sink_w is not None:
                sum_rows(part, out=sink_w.view(-1), accumulate=True)
            else:
                gw = sum_rows(part).view(ctx.weight.shape).to(ctx.weight.dtype)
        if need_b:
            if sink_b is not None:
                sum_rows(g2, out=sink_b, accumulate=True)
            else:
                gb = sum_rows(g2).to(ctx.bias.dtype)
        return route((gx, gw, gb, None, None))

    @staticmethod
    def _backward_scaled(ctx, gy, x, w_c, rowscale, want_x):
        """Backward of the branch y = res + diag(rowscale) (x W^T + b) on the kernels that take the scale as an argument;
        None if one of them declines (the caller then scales the gradient and takes the general path)."""
        N, K = w_c.shape
        if not (mfma.ENABLED and w_c.dtype != torch.float32):
            return None
        if gy.dtype != w_c.dtype:
            gy = gy.to(w_c.dtype)
        g2 = gy.reshape(-1, N)
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        need_w, need_b = ctx.needs_input_grad[1], ctx.bias is not None and ctx.needs_input_grad[2]
        sink_w, sink_b = grad_sink(ctx.weight), grad_sink(ctx.bias)
        if (need_w and sink_w is None) or (need_b and sink_b is None) or (need_b and not need_w):
            return None
        gx = None
        if want_x:
            if ctx.z is not None:
                gx = mfma.gemm_nt(g2, transposed(ctx.weight, w_c.dtype), res=ctx.z.reshape(-1, K), rowscale=rowscale,
                                  rows_per_sample=ctx.rps, act=4)
            else:
                gx = mfma.gemm_nt(g2, transposed(ctx.weight, w_c.dtype), rowscale=rowscale, rows_per_sample=ctx.rps)
            if gx is None:
                return None
            gx = gx.view(x.shape)
        if need_w:
            if not mfma.defer_gemm_tn(g2, x.reshape(-1, K), sink_w, sink_b if need_b else None, rowscale, ctx.rps) and \
                    mfma.gemm_tn(g2, x.reshape(-1, K), out=sink_w, bias_out=sink_b if need_b else None, rowscale=rowscale,
                                 rows_per_sample=ctx.rps) is None:
                if gx is not None:
                    return None                          # (nothing accumulated yet: gemm_tn declined before launching)
                return None
        return gx, None, None, None, None


def linear_tokens(x2, weight, bias, cd):
    """x2 (T, K) @ weight^T + bias for a weight parameter of shape (N, K) or (N, K, 1, 1) (a 1x1 convolution on
    channels-last tokens), on the MFMA GEMM kernels: forward, input gradient and parameter gradients (accumulated into
    the flat gradient buffer).  `cd`: 16-bit compute dtype."""
    w_c = as_dtype(weight, cd)
    w_c = w_c.view(w_c.shape[0], -1)
    b_c = as_dtype(bias, cd)
    if torch.is_grad_enabled() and (weight.requires_grad or x2.requires_grad):
        return _LinearFn.apply(x2, weight, bias, w_c, b_c, None, None, None)
    y = mfma.gemm_nt(x2, w_c, b_c)
    if y is None:
        mfma.note_library("linear.fwd", x2, w_c)
    return F.linear(x2, w_c, b_c) if y is None else y


class Linear(nn.Linear):
    def forward(self, x, res=None, rowscale=None, z=None):
        """y = x W^T + b.  `z`: see _LinearFn (x = gelu(z), gradient routed to z).  Gradient-free callers may pass `res` (same shape as y) and a per-sample fp32 `rowscale`
        (B,): y = res + rowscale[b] * (x W^T + b), the stochastic-depth residual of mix_transformer.py:203-207 fused
        into the GEMM epilogue."""
        if not x.is_cuda:
            y = F.linear(x, self.weight, self.bias)
            return y if res is None else _residual(res, y, rowscale)
        cd = compute_dtype(x)
        if cd == torch.float32:
            from . import split32
            if split32.usable(x):                      # fp32 parity mode: split-bf16 products on the matrix-core kernels
                y = split32.linear(x.float(), self.weight, self.bias)
                return y if res is None else _residual(res, y, rowscale)
        w_c, b_c = as_dtype(self.weight, cd), as_dtype(self.bias, cd)
        if x.dtype != cd:
            x = x.to(cd)
        if torch.is_grad_enabled() and (self.weight.requires_grad or x.requires_grad):
            if res is not None and _FUSED_RESIDUAL and mfma.ENABLED and cd != torch.float32 and res.dtype == cd and \
                    res.is_contiguous() and x.is_contiguous() and w_c.shape[1] % 64 == 0 and w_c.shape[0] % 64 == 0 and \
                    grad_sink(self.weight) is not None and (self.bias is None or grad_sink(self.bias) is not None):
                return _LinearFn.apply(x, self.weight, self.bias, w_c, b_c, res, rowscale, z)
            y = _LinearFn.apply(x, self.weight, self.bias, w_c, b_c, None, None, z)
            return y if res is None else _residual(res, y, rowscale)
        N, K = w_c.shape
        x2 = x.reshape(-1, K)
        if res is not None and res.dtype == cd and res.is_contiguous():
            y = mfma.gemm_nt(x2, w_c, b_c, res=res.view(-1, N), rowscale=rowscale,
                             rows_per_sample=(x2.shape[0] // x.shape[0]) if rowscale is not None else 0)
            if y is not None:
                return y.view(x.shape[:-1] + (N,))
        y = mfma.gemm_nt(x2, w_c, b_c)
        if y is None:
            mfma.note_library("linear.fwd", x2, w_c)
        y = F.linear(x, w_c, b_c) if y is None else y.view(x.shape[:-1] + (N,))
        return y if res is None else _residual(res, y, rowscale)


def _residual(res, y, rowscale):
    if rowscale is None:
        return res + y
    return torch.addcmul(res, y, rowscale.to(y.dtype).view((-1,) + (1,) * (y.dim() - 1)))
