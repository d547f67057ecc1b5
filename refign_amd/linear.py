"""nn.Linear with a weight-gradient GEMM shaped for MI355X.

The weight gradient of a token-wise Linear is dW[N,K] = dY[T,N]^T . X[T,K] with T = tokens (8 160 ... 129 600 on the
Refign step) and N, K <= 2048: a "small output, very long reduction" GEMM.  The library picks a 64x64 macro-tile without
split-K for it, i.e. 25-400 workgroups each walking the whole T: measured 24 TF/s on [320 x 8160] x [8160 x 320]
(240 of those per step, 58 ms of wgrad GEMMs in total, profiles/r01_step_shapes_bf16_findnormal.txt).  Splitting T into
S independent slabs turns it into a batched GEMM with S x more workgroups plus one (S, N, K) fp32 reduction.
Forward and the input gradient stay ordinary library GEMMs.  Same parameters / state_dict keys as nn.Linear.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _split(T):
    for s in (64, 32, 16, 8, 4, 2):
        if T % s == 0 and T // s >= 384:
            return s
    return 1


class _LinearFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.bfloat16)
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(x, w, b)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        N, K = w.shape
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.matmul(gy, w)
        g2 = gy.reshape(-1, N)
        if ctx.needs_input_grad[1]:
            x2 = x.reshape(-1, K)
            T = x2.shape[0]
            S = _split(T)
            if S > 1:
                part = torch.bmm(g2.view(S, T // S, N).transpose(1, 2), x2.view(S, T // S, K))     # (S, N, K)
                gw = part.sum(0, dtype=torch.float32)
            else:
                gw = g2.t().mm(x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0, dtype=torch.float32)
        return gx, gw, gb


class Linear(nn.Linear):
    def forward(self, x):
        if x.is_cuda and torch.is_grad_enabled() and self.weight.requires_grad:
            return _LinearFn.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)
