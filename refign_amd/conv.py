"""nn.Conv2d for the MiT patch-embedding / spatial-reduction convolutions, with the parameter plumbing of params.py.

On a GPU the convolution runs on the hand-written kernels: the implicit-GEMM MFMA kernel (conv2d_mfma: forward; conv2d_mfma_grad:
forward + data gradient + weight gradient under autograd), the patch GEMM for kernel == stride (patch_conv_tokens), split-bf16
products for fp32 tensors (split32.conv2d).  Weights are used through cached 16-bit copies (params.derived) refreshed in place after
optimiser / EMA updates; weight / bias gradients are added straight into the fp32 views of the flat gradient buffer.  What is left
to ATen: CPU tensors (host-side tests) and geometries outside the kernels' domain (unequal strides / paddings, grouped
convolutions other than depthwise) -- each such GPU call is recorded by mfma.note_library.
Same parameters / state_dict keys as nn.Conv2d.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import mfma as _mfma
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
        if cd == torch.float32 and self.groups == 1:
            # fp32 parity mode: split-bf16 products on the matrix-core kernels (refign_amd/split32.py)
            from . import split32
            if split32.usable(x):
                y = split32.conv2d(x.float(), self.weight, self.bias, self.stride, self.padding, self.dilation)
                if y is not None:
                    return y
        # the MiT token tensors reach the convolutions as channels-last views, and the library then wants the filter in
        # channels-last too: keep the cached copy in that layout instead of converting it at every call
        w_c = derived(self.weight, (cd, "channels_last"),
                      lambda t: t.to(cd).contiguous(memory_format=torch.channels_last), _identity)
        b_c = as_dtype(self.bias, cd)
        if x.dtype != cd:
            x = x.to(cd)
        if torch.is_grad_enabled() and (self.weight.requires_grad or x.requires_grad):
            if self.groups == 1:
                y = conv2d_mfma_grad(x, self.weight, self.bias, self.stride, self.padding, self.dilation, cd)
                if y is not None:
                    return y
            _mfma.note_library("conv2d.autograd", x, self.weight)
            return _Conv2dFn.apply(x, self.weight, self.bias, w_c, b_c, self.stride, self.padding, self.dilation,
                                   self.groups)
        if self.groups == 1 and not torch.is_grad_enabled():
            y = conv2d_mfma(x, self.weight, self.bias, self.stride, self.padding, self.dilation, dtype=cd)
            if y is not None:
                return y
        _mfma.note_library("conv2d", x, self.weight)
        return F.conv2d(x, w_c, b_c, self.stride, self.padding, self.dilation, self.groups)


# ---------------------------------------------------------------------------------------------------------------------
# Spatial-reduction convolution of the MiT attention (kernel = stride = sr_ratio, no padding: mix_transformer.py:133-146)
# as what it is -- a Linear over non-overlapping r x r patches -- on the token layout, without the NCHW round trip.
# Forward: one gather copy (tokens -> patches) + one GEMM with bias, output directly as (B, N', C) tokens for the
# LayerNorm that follows.  Backward: the Linear's three GEMM-shaped ops (split-T weight gradient, parameter gradients
# accumulated into the flat buffer) + one scatter copy; the library's convolution backward for this shape is five
# launches of its own plus per-call zero-fill / cast tensor ops (~85 us of GPU time per layer per pass, 240 per step).
# ---------------------------------------------------------------------------------------------------------------------
_DT = {torch.float32: 0, torch.bfloat16: 1}


def _split32():
    from . import split32
    return split32


def _split_ok(*ts):
    return all(t.dtype == torch.float32 for t in ts) and _split32().usable(*ts)


# Patch rows in (c, ry, rx) order -- the layout of the convolution weight itself (csrc/upcat.hip patchify_cmajor_kernel):
# the patch GEMM uses the plain 16-bit copy of the parameter, the input-gradient GEMM its plain transpose, and the weight
# gradient is added straight into the parameter's .grad by the TN kernel (bias gradient in the same launch) instead of
# partial sums + a reduction + a permuted add + two bias-reduction launches.  (Module constant; False: (ry, rx, c) rows.)
_PATCH_CMAJOR = True


def _cmajor_ok(x, C, r):
    return _PATCH_CMAJOR and r in (2, 4, 8) and x.is_cuda and x.dtype in _DT and C % 8 == 0 and C >= 16 and \
        r * r * C * x.element_size() <= 65528


def _patchify(src, dst, B, H, W, C, r, inverse, cmajor=False):
    lib = _lib.load_library()
    with on_device(src.device):
        fn = lib.rfn_patchify_tokens_cmajor if cmajor else lib.rfn_patchify_tokens
        rc = fn(ptr(src), ptr(dst), B, H, W, C, r, _DT[src.dtype], 1 if inverse else 0, current_stream(src.device))
    _lib.check(rc, "patchify_tokens")


def _to_patches(x, H, W, r, cmajor=False):
    B, N, C = x.shape
    Hr, Wr = H // r, W // r
    if x.is_cuda and x.dtype in _DT and C % 8 == 0 and x.is_contiguous():
        out = torch.empty((B * Hr * Wr, r * r * C), dtype=x.dtype, device=x.device)
        _patchify(x, out, B, H, W, C, r, False, cmajor)              # one vectorised gather (csrc/upcat.hip)
        return out, Hr, Wr
    v = x.view(B, H, W, C)
    if Hr * r != H or Wr * r != W:
        v = v[:, :Hr * r, :Wr * r]                   # the strided conv drops the ragged border
    order = (0, 1, 3, 5, 2, 4) if cmajor else (0, 1, 3, 2, 4, 5)
    return v.reshape(B, Hr, r, Wr, r, C).permute(*order).reshape(B * Hr * Wr, r * r * C), Hr, Wr


def _from_patches(gp, B, H, W, C, r, Hr, Wr, cmajor=False):
    if gp.is_cuda and gp.dtype in _DT and C % 8 == 0 and gp.is_contiguous():
        ragged = Hr * r != H or Wr * r != W
        out = (torch.zeros if ragged else torch.empty)((B, H * W, C), dtype=gp.dtype, device=gp.device)
        _patchify(gp, out, B, H, W, C, r, True, cmajor)
        return out
    if cmajor:
        gp = gp.view(B, Hr, Wr, C, r, r).permute(0, 1, 2, 4, 5, 3)
    g = gp.reshape(B, Hr, Wr, r, r, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hr * r, Wr * r, C)
    if Hr * r != H or Wr * r != W:
        g = F.pad(g, (0, 0, 0, W - Wr * r, 0, H - Hr * r))
    return g.reshape(B, H * W, C)


def _krsc_view(p):
    return p.permute(2, 3, 1, 0)


class _PatchLinearFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias, w2, b_c, H, W, r, cmajor=False):
        patches, Hr, Wr = _to_patches(x, H, W, r, cmajor)
        ctx.save_for_backward(patches, w2)
        ctx.weight, ctx.bias, ctx.cmajor = weight, bias, cmajor
        ctx.geom = (x.shape, H, W, r, Hr, Wr)
        y = _mfma.gemm_nt(patches, w2, b_c)
        if y is None and _split_ok(patches, w2):
            y = _split32().gemm_nt(patches, w2, b_c)       # fp32 parity mode (refign_amd/split32.py)
        if y is None:
            _mfma.note_library("patch_linear.fwd", patches, w2)
        return (F.linear(patches, w2, b_c) if y is None else y).view(x.shape[0], Hr * Wr, w2.shape[0])

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
        cmajor = ctx.cmajor
        if ctx.needs_input_grad[0]:
            # (K, Co) = W^T of the patch Linear: dx = dy W on the NT kernel.  Cached as its (r, r, C, Co) view, the shape of
            # weight.permute(2, 3, 1, 0), so that params.refresh() re-fills it in place (graph replays see live weights);
            # channel-major patches: the plain transpose of the (Co, C r r) parameter matrix
            if cmajor:
                from .params import transposed
                wT = transposed(ctx.weight, w2.dtype)
            else:
                wT = derived(ctx.weight, (w2.dtype, "patch_linear_T"),
                             lambda t: t.to(w2.dtype).permute(2, 3, 1, 0).contiguous(), _krsc_view).view(K, Co)
            gp = _mfma.gemm_nt(g2, wT)
            if gp is None and _split_ok(g2, wT):
                gp = _split32().gemm_nt(g2, wT)
            if gp is None:
                _mfma.note_library("patch_linear.dgrad", g2, w2)
            gx = _from_patches(torch.mm(g2, w2) if gp is None else gp, B, H, W, C, r, Hr, Wr, cmajor)
        T = g2.shape[0]
        S = _split(T)
        sw, sb = grad_sink(ctx.weight), grad_sink(ctx.bias)
        if cmajor and sw is not None and (ctx.bias is None or sb is not None) and \
                _mfma.gemm_tn(g2, patches, out=sw.view(Co, K), bias_out=sb) is not None:
            return gx, None, None, None, None, None, None, None, None      # both gradients added in that one launch
        part = _mfma.gemm_tn(g2, patches)                                               # fp32 slab partials (S, Co, K)
        if part is None and _split_ok(g2, patches):
            part = _split32().gemm_tn(g2, patches)[None]
        if part is None:
            _mfma.note_library("patch_linear.wgrad", g2, patches)
        if part is not None:
            part = part.view(part.shape[0], Co * K)
        elif S > 1:
            part = torch.bmm(g2.view(S, T // S, Co).transpose(1, 2), patches.view(S, T // S, K)).view(S, Co * K)
        else:
            part = g2.t().mm(patches).view(1, Co * K)
        # the parameter is (Co, C, r, r), the GEMM's weight is its (Co, r, r, C) permutation (channel-major patches: itself)
        gw2 = sum_rows(part).view((Co, C, r, r) if cmajor else (Co, r, r, C))
        if sw is not None:
            (sw if cmajor else sw.permute(0, 2, 3, 1)).add_(gw2)
        else:
            gw = (gw2 if cmajor else gw2.permute(0, 3, 1, 2)).to(ctx.weight.dtype)
        if ctx.bias is not None:
            if sb is not None:
                sum_rows(g2, out=sb, accumulate=True)
            else:
                gb = sum_rows(g2).to(ctx.bias.dtype)
        return gx, gw, gb, None, None, None, None, None, None


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
    # channel-major patches where the weight gradient is wanted (the student); the gradient-free networks keep the (ry, rx, c)
    # rows, whose gather needs no transposition (teacher, 40 views: 16 vs 22 us per call)
    needs_grad = torch.is_grad_enabled() and conv.weight.requires_grad
    cmajor = needs_grad and _cmajor_ok(x, C, r) and cd in _DT
    if cmajor:
        w2 = as_dtype(conv.weight, cd).detach().view(Co, C * r * r)
    else:
        w2 = derived(conv.weight, (cd, "patch_linear"), lambda t: t.to(cd).permute(0, 2, 3, 1).contiguous(),
                     lambda t: t.permute(0, 2, 3, 1)).view(Co, r * r * C)
    b_c = as_dtype(conv.bias, cd)
    if x.dtype != cd:
        x = x.to(cd)
    if needs_grad:
        return _PatchLinearFn.apply(x, conv.weight, conv.bias, w2, b_c, H, W, r, cmajor)
    patches, Hr, Wr = _to_patches(x, H, W, r, cmajor)
    y = _mfma.gemm_nt(patches, w2, b_c)
    if y is None and _split_ok(patches, w2):
        y = _split32().gemm_nt(patches, w2, b_c)
    if y is None:
        _mfma.note_library("patch_linear.fwd", patches, w2)
    return (F.linear(patches, w2, b_c) if y is None else y).view(x.shape[0], Hr * Wr, Co)


# ---------------------------------------------------------------------------------------------------------------------
# Dense convolution on the hand-written implicit-GEMM MFMA kernel (csrc/mfma_gemm.hip: rfn_conv2d_nhwc), for the
# gradient-free networks of the step: the frozen matcher (VGG-16, flow decoders, refinement, uncertainty tail --
# vgg.py:108-120, modules.py:395-561) under its fp16 autocast and the EMA-teacher / ImageNet MiT patch embeddings.
# Tensors keep their NCHW SHAPE and carry channels-last strides between layers, so consecutive layers hand each other
# NHWC memory without a copy; bias, folded BatchNorm and ReLU / LeakyReLU ride in the epilogue.
# ---------------------------------------------------------------------------------------------------------------------
import os as _os

_ACT = {None: 0, 'relu': 1, 'leaky': 3}
_CONV_MFMA = True


def _nhwc_view(p):
    return p.permute(0, 2, 3, 1)


def _packed(weight, bias, dtype, n_mult=8):
    """Cached ([Np, Kpad] tap-major 16-bit weight with the output channels padded to `n_mult`, bias padded likewise).  What is
    cached per parameter is the (N, KH, KW, C) VIEW into the padded buffer -- the same shape as `weight.permute(0, 2, 3, 1)`
    -- so that params.refresh() re-fills it IN PLACE after an optimizer / EMA update: a captured graph keeps pointing
    at live weights (a re-made copy would leave the replay with stale ones)."""
    def make(t):
        N, C, KH, KW = t.shape
        N8, Cp = -(-N // n_mult) * n_mult, -(-C // 8) * 8
        Kp = -(-(KH * KW * Cp) // 64) * 64
        base = torch.zeros((N8, Kp), dtype=dtype, device=t.device)
        view = base[:N, :KH * KW * Cp].view(N, KH, KW, Cp)[..., :C]
        view.copy_(t.permute(0, 2, 3, 1))
        view._rfn_base = base
        return view
    wp = derived(weight, ("igemm", dtype, n_mult), make, _nhwc_view)._rfn_base
    bp = None
    if bias is not None:
        def makeb(t):
            base = torch.zeros(-(-t.shape[0] // n_mult) * n_mult, dtype=dtype, device=t.device)
            view = base[:t.shape[0]]
            view.copy_(t)
            view._rfn_base = base
            return view
        bp = derived(bias, ("igemm_bias", dtype, n_mult), makeb, _identity)._rfn_base
    return wp, bp


def _cnhw_view(p):
    return p.permute(1, 2, 3, 0)


def _packed_t(weight, dtype, n_mult):
    """The filter of the DATA-gradient product: rows c (padded to 8), columns [tap][n] with n padded to `n_mult` like the
    forward's output channels -- Wt of rfn_conv2d_nhwc_dgrad.  Cached / re-filled in place like _packed."""
    def make(t):
        N, C, KH, KW = t.shape
        Np, Cp = -(-N // n_mult) * n_mult, -(-C // 8) * 8
        Kp = -(-(KH * KW * Np) // 64) * 64
        base = torch.zeros((Cp, Kp), dtype=dtype, device=t.device)
        view = base[:C, :KH * KW * Np].view(C, KH, KW, Np)[..., :N]
        view.copy_(t.permute(1, 2, 3, 0))
        view._rfn_base = base
        return view
    return derived(weight, ("igemm_T", dtype, n_mult), make, _cnhw_view)._rfn_base


def _nhwc16(x, dtype, Cp):
    """NCHW-shaped tensor (any strides) -> contiguous channels-last (B, H, W, Cp) in `dtype`, channels zero-padded to Cp."""
    xh = x.permute(0, 2, 3, 1)
    C = xh.shape[-1]
    if Cp != C:
        buf = torch.zeros(xh.shape[:3] + (Cp,), dtype=dtype, device=x.device)
        buf[..., :C] = xh
        return buf
    if xh.dtype != dtype or not xh.is_contiguous():
        xh = xh.to(dtype).contiguous()
    return xh


class _ConvMfmaFn(torch.autograd.Function):
    """Dense convolution (groups 1, square kernel / stride / padding / dilation) with forward, data gradient and weight
    gradient on the hand-written matrix-core kernels: implicit GEMM (csrc/mfma_gemm.hip: rfn_conv2d_nhwc), the same kernel
    in transposed-gather mode (rfn_conv2d_nhwc_dgrad) and the split-T weight-gradient kernel with gathered im2col rows
    (rfn_conv2d_nhwc_wgrad).  The trainable convolutions of the student: DAFormer 3x3 bottleneck (daformer.py:65-126), MiT
    overlap patch embeddings (mix_transformer.py:210-242), the 19-class 1x1s (output channels padded to 64 inside)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias, s, p, d, dtype, act):
        N, C, KH, KW = weight.shape
        Cp, Np = -(-C // 8) * 8, -(-N // 64) * 64
        wp, bp = _packed(weight, bias, dtype, 64)
        xh = _nhwc16(x, dtype, Cp)
        y = _mfma.conv2d_nhwc(xh, wp, bp, KH, KW, s, p, d, act=0)
        if y is None:
            raise RuntimeError("conv2d (MFMA, autograd): operands outside the kernel's domain")
        ctx.save_for_backward(xh)
        ctx.weight, ctx.bias, ctx.conf, ctx.dtype = weight, bias, (s, p, d), dtype
        ctx.xshape = x.shape
        return (y[..., :N] if Np != N else y).permute(0, 3, 1, 2)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        from .params import sum_rows
        (xh,) = ctx.saved_tensors
        weight, bias, (s, p, d), dtype = ctx.weight, ctx.bias, ctx.conf, ctx.dtype
        N, C, KH, KW = weight.shape
        B, H, W, Cp = xh.shape
        Np = -(-N // 64) * 64
        gh = _nhwc16(gy, dtype, Np)                                  # (B, OH, OW, Np), zero in the padded channels
        gx = gw = gb = None
        need_w = ctx.needs_input_grad[1]
        need_b = bias is not None and ctx.needs_input_grad[2]

        def library_backward(want_x, want_w, want_b):
            """a shape outside a backward kernel's domain (conv2d_mfma_grad pre-checks the Python-side limits only): the
            library's convolution_backward for the missing gradients, recorded like every other library call"""
            _mfma.note_library("convolution_backward", gy, weight)
            xn = xh[..., :C].permute(0, 3, 1, 2)
            return torch.ops.aten.convolution_backward(
                gy.to(dtype), xn, weight.to(dtype), [N] if bias is not None else None, [s, s], [p, p], [d, d], False, [0, 0], 1,
                [want_x, want_w, want_b])

        if ctx.needs_input_grad[0]:
            dx = _mfma.conv2d_nhwc_dgrad(gh, _packed_t(weight, dtype, 64), H, W, Cp, KH, KW, s, p, d)
            if dx is None:
                gx = library_backward(True, False, False)[0]
            else:
                gx = (dx[..., :C] if Cp != C else dx).permute(0, 3, 1, 2)
        if need_w or need_b:
            sw, sb = grad_sink(weight), grad_sink(bias)
            Kp = -(-(KH * KW * Cp) // 64) * 64
            bsum = torch.zeros(Np, dtype=torch.float32, device=gh.device) if need_b else None
            part = _mfma.conv2d_nhwc_wgrad(gh, xh, KH, KW, Kp, s, p, d, bias_out=bsum)
            if part is None:
                _, lw, lb = library_backward(False, need_w, need_b)
                if need_w:
                    if sw is not None:
                        sw.add_(lw.to(sw.dtype))
                    else:
                        gw = lw.to(weight.dtype)
                if need_b:
                    if sb is not None:
                        sb.add_(lb.to(sb.dtype))
                    else:
                        gb = lb.to(bias.dtype)
                return gx, gw, gb, None, None, None, None, None
            if need_w:
                S = part.shape[0]
                g2 = sum_rows(part.view(S, Np * Kp)).view(Np, Kp)[:N, :KH * KW * Cp].view(N, KH, KW, Cp)[..., :C]
                if sw is not None:
                    sw.permute(0, 2, 3, 1).add_(g2)                  # the parameter is (N, C, KH, KW)
                else:
                    gw = g2.permute(0, 3, 1, 2).to(weight.dtype)
            if need_b:
                if sb is not None:
                    sb.add_(bsum[:N])
                else:
                    gb = bsum[:N].to(bias.dtype)
        return gx, gw, gb, None, None, None, None, None


def conv2d_mfma_grad(x, weight, bias, stride, padding, dilation, dtype):
    """conv2d under autograd on the hand-written kernels (see _ConvMfmaFn); None outside their domain (caller takes the
    library path and records it)."""
    if not (_CONV_MFMA and _mfma.ENABLED and x.is_cuda and x.dim() == 4 and dtype in (torch.float16, torch.bfloat16)):
        return None
    for v in (stride, padding, dilation):
        if isinstance(v, (tuple, list)) and v[0] != v[1]:
            return None
    s, p, d = (v[0] if isinstance(v, (tuple, list)) else v for v in (stride, padding, dilation))
    N, C, KH, KW = weight.shape
    Cp = -(-C // 8) * 8
    if x.shape[1] != C or s & (s - 1) or isinstance(p, str) or -(-(KH * KW * Cp) // 64) * 64 >= 65536 \
            or KH * KW * (-(-N // 64) * 64) // 8 >= 65536:
        return None
    return _ConvMfmaFn.apply(x, weight, bias, s, p, d, dtype, None)


def conv2d_mfma(x, weight, bias, stride=1, padding=0, dilation=1, act=None, dtype=None):
    """F.conv2d(x, weight, bias, stride, padding, dilation) [+ ReLU / LeakyReLU(0.1)] for groups == 1 on the GPU, 16-bit
    operands with fp32 accumulation.  x: (B, C, H, W) in any memory format (channels-last is used in place); returns a
    (B, N, OH, OW) tensor with channels-last strides, or None when the call is outside the kernel's domain (caller falls
    back to the library).  Gradient-free only."""
    if not (_CONV_MFMA and _mfma.ENABLED and x.is_cuda and x.dim() == 4 and not torch.is_grad_enabled()):
        return None
    dtype = dtype or x.dtype
    if dtype not in (torch.float16, torch.bfloat16):
        return None
    s, p, d = (v[0] if isinstance(v, (tuple, list)) else v for v in (stride, padding, dilation))
    for v in (stride, padding, dilation):
        if isinstance(v, (tuple, list)) and v[0] != v[1]:
            return None
    N, C, KH, KW = weight.shape
    if x.shape[1] != C:
        return None
    wp, bp = _packed(weight, bias, dtype)
    xh = x.permute(0, 2, 3, 1)
    Cp = -(-C // 8) * 8
    if xh.dtype != dtype or Cp != C or not xh.is_contiguous():
        if Cp != C:
            buf = torch.zeros(xh.shape[:3] + (Cp,), dtype=dtype, device=x.device)
            buf[..., :C] = xh
            xh = buf
        else:
            xh = xh.to(dtype).contiguous()
    y = _mfma.conv2d_nhwc(xh, wp, bp, KH, KW, s, p, d, act=_ACT[act])
    if y is None:
        return None
    if y.shape[-1] != N:
        y = y[..., :N]
    return y.permute(0, 3, 1, 2)
