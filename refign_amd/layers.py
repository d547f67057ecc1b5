"""Shared layer containers with the reference's parameter names (models/modules.py:16-68,564-596).

These hold parameters under the same state_dict keys as the reference (`conv.weight`, `bn.running_mean`,
`depthwise_conv.conv.weight`, `pointwise_conv.bn.bias`, `proj.weight`) and execute either the training formulation
(conv -> BatchNorm with batch statistics -> activation) or, for frozen/eval networks under no_grad, a folded one
(BatchNorm merged into the conv weights once, activation applied in place on the conv output).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

LEAKY_SLOPE = 0.1   # activation_layer = partial(nn.LeakyReLU, negative_slope=0.1)  (modules.py:407,459,498)


_DW_BN_STATS = True
_DW_BN_FUSED = True


class ConvBNReLU(nn.Module):
    """Constructor semantics and parameter names of models/modules.py:16-56, including the depthwise-separable form
    (a depthwise ConvBNReLU followed by a 1x1 pointwise ConvBNReLU, each with its own norm + activation)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, groups=1, padding=None,
                 norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU, bias='auto', depthwise_separable=False,
                 inplace=True, affine=True):
        super().__init__()
        padding = dilation * (kernel_size - 1) // 2 if padding is None else padding
        self.use_norm = norm_layer is not None
        self.use_activation = activation_layer is not None
        self.depthwise_separable = depthwise_separable
        if bias == 'auto':
            bias = not self.use_norm
        if depthwise_separable:
            assert kernel_size > 1 and groups == 1
            self.depthwise_conv = ConvBNReLU(in_channels, in_channels, kernel_size, stride=stride, padding=padding,
                                             dilation=dilation, groups=in_channels, norm_layer=norm_layer,
                                             activation_layer=activation_layer)
            self.pointwise_conv = ConvBNReLU(in_channels, out_channels, 1, norm_layer=norm_layer,
                                             activation_layer=activation_layer)
        else:
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation=dilation,
                                  groups=groups, bias=bias)
            if self.use_norm:
                self.bn = norm_layer(out_channels, affine=affine)
        self.act, self.act_slope = None, 0.0
        if self.use_activation:
            probe = activation_layer()
            self.act = 'leaky' if isinstance(probe, nn.LeakyReLU) else 'relu'
            self.act_slope = getattr(probe, 'negative_slope', 0.0)
        self._folded = None

    # -- eval-time folding -------------------------------------------------------------------------------------------
    def folded(self):
        """(weight, bias) with eval-mode BatchNorm folded in; cached until train() / load_state_dict()."""
        if self._folded is None:
            w, b = self.conv.weight, self.conv.bias
            if self.use_norm:
                bn = self.bn
                inv = torch.rsqrt(bn.running_var + bn.eps)
                g = inv if bn.weight is None else bn.weight * inv
                w = w * g.view(-1, 1, 1, 1)
                shift = -bn.running_mean * g
                if bn.bias is not None:
                    shift = shift + bn.bias
                b = shift if b is None else b * g + shift
            self._folded = (w.detach().contiguous(), None if b is None else b.detach().contiguous())
        return self._folded

    def train(self, mode=True):
        self._folded = None
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._folded = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *a, **k):
        self._folded = None
        return super()._apply(fn, *a, **k)

    def _out_count(self, x):
        """values per channel of this block's convolution result on `x` (batch x OH x OW)"""
        c = self.conv
        n = x.shape[0]
        for i in (0, 1):
            pad = c.padding[i] if not isinstance(c.padding, str) else (c.dilation[i] * (c.kernel_size[i] - 1) // 2 if c.padding == "same" else 0)
            n *= max(0, (x.shape[2 + i] + 2 * pad - c.dilation[i] * (c.kernel_size[i] - 1) - 1) // c.stride[i] + 1)
        return n

    def _conv2d(self, x, w, b, stats=None):
        """The convolution itself.  Depthwise 3x3 (the DAFormer ASPP branches) goes to the hand-written channels-last
        HIP kernel on the GPU -- the library's grouped-conv path for group size 1 is ~50x off the HBM roofline; every
        other shape is a dense conv on the ROCm library."""
        c = self.conv
        if (x.is_cuda and c.groups == c.in_channels == c.out_channels and c.kernel_size == (3, 3)
                and c.stride == (1, 1) and c.padding == c.dilation and c.dilation[0] == c.dilation[1]
                and c.in_channels % 8 == 0):
            from .dwconv import dwconv3x3_nhwc
            xh = x.permute(0, 2, 3, 1)                       # free for channels_last inputs
            y = dwconv3x3_nhwc(xh if xh.is_contiguous() else xh.contiguous(), w, b, c.dilation[0], stats=stats)
            return y.permute(0, 3, 1, 2)                     # NCHW-shaped, channels_last strides
        if stats is not None:
            raise RuntimeError("ConvBNReLU._conv2d(stats=...): depthwise 3x3 on the HIP kernel only")
        if x.is_cuda:
            from . import mfma, split32
            if c.groups == 1 and x.dtype == torch.float32 and not torch.is_autocast_enabled("cuda") and split32.usable(x):
                y = split32.conv2d(x, w, b, c.stride, c.padding, c.dilation)     # fp32 parity mode (split-bf16 products)
                if y is not None:
                    return y
            from .params import compute_dtype
            cd = compute_dtype(x)
            if c.groups == 1 and cd in (torch.float16, torch.bfloat16):
                # blocks without a norm layer / with a BatchNorm the fused kernel does not take (the matcher's decoders,
                # refinement and uncertainty networks under fp16 autocast -- alignment_model.py:81-146):
                # forward, data and weight gradient on the implicit-GEMM kernels, as the student's 3x3 bottleneck
                from .conv import conv2d_mfma, conv2d_mfma_grad
                if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
                    y = conv2d_mfma_grad(x, w, b, c.stride, c.padding, c.dilation, cd)
                else:
                    y = conv2d_mfma(x, w, b, c.stride, c.padding, c.dilation, dtype=cd)
                if y is not None:
                    return y
            mfma.note_library("conv2d.autograd" if torch.is_grad_enabled() else "conv2d", x, w)
        return F.conv2d(x, w, b, c.stride, c.padding, c.dilation, c.groups)

    def _conv_train(self, x, cd):
        """The convolution of a training-mode block on the GPU in the 16-bit compute dtype `cd`.  1x1 (the ASPP / fusion
        projections): a Linear over channels-last tokens on the MFMA GEMM kernels, forward and both gradients; depthwise
        3x3: the HIP stencil; dense k x k without autograd (EMA teacher): the implicit-GEMM kernel; dense k x k under
        autograd (the student's 3x3 bottleneck): the library convolution."""
        c = self.conv
        # (under autograd the Linear's gradient GEMMs want N % 64 == 0; other widths take the implicit-GEMM path below, which
        # pads its output channels itself)
        if c.groups == 1 and c.kernel_size == (1, 1) and c.stride == (1, 1) and c.padding == (0, 0) and \
                c.in_channels % 64 == 0 and c.out_channels % (64 if torch.is_grad_enabled() else 8) == 0:
            from .linear import linear_tokens
            xh = x.permute(0, 2, 3, 1)
            if xh.dtype != cd:
                xh = xh.to(cd)
            if not xh.is_contiguous():
                xh = xh.contiguous()
            B, H, W, C = xh.shape
            y = linear_tokens(xh.reshape(B * H * W, C), c.weight, c.bias, cd)
            return y.view(B, H, W, -1).permute(0, 3, 1, 2)
        if c.groups == 1 and not torch.is_grad_enabled():
            from .conv import conv2d_mfma
            y = conv2d_mfma(x, c.weight, c.bias, c.stride, c.padding, c.dilation, dtype=cd)
            if y is not None:
                return y
        if c.groups == 1 and torch.is_grad_enabled():
            # dense k x k under autograd (the student's 3x3 bottleneck): forward, data and weight gradient on the implicit-GEMM
            # kernels (conv._ConvMfmaFn)
            from .conv import conv2d_mfma_grad
            y = conv2d_mfma_grad(x, c.weight, c.bias, c.stride, c.padding, c.dilation, cd)
            if y is not None:
                return y
        return self._conv2d(x, c.weight, c.bias)

    def fused_out_ok(self, x):
        """forward(x, out=...) is available: gradient-free, training-mode BatchNorm on the fused kernel (the EMA teacher's
        decode head) -- the block's result can be written straight into a channel slice of a wider tensor."""
        m = self.pointwise_conv if self.depthwise_separable else self
        if not (x.is_cuda and not torch.is_grad_enabled() and m.use_norm and m.training and m.act in (None, 'relu')
                and os.environ.get("RFN_BN_KERNEL", "1") != "0"):
            return False
        from . import bn as bnk
        from .params import compute_dtype
        return bnk.usable(x, m.bn, compute_dtype(x), channels=m.conv.out_channels, count=m._out_count(x))

    def _dw_stats_ok(self, x, cd):
        """depthwise 3x3 -> BatchNorm(train): the convolution kernel leaves the batch statistics of its result behind
        (csrc/dwconv.hip STATS), the BatchNorm skips its statistics pass.  (_DW_BN_STATS = False: two passes.)"""
        c = self.conv
        return (_DW_BN_STATS and x.is_cuda and x.dtype == torch.bfloat16 and cd == torch.bfloat16
                and c.groups == c.in_channels == c.out_channels and c.kernel_size == (3, 3) and c.stride == (1, 1)
                and c.padding == c.dilation and c.dilation[0] == c.dilation[1] and c.in_channels % 8 == 0)

    def _bn_train(self, x, cd, out=None):
        from . import bn as bnk
        act = {None: 0, 'relu': 1, 'leaky': 3}[self.act]
        if self._dw_stats_ok(x, cd) and not torch.is_grad_enabled() and out is None and act in (0, 1) and _DW_BN_FUSED \
                and self.bn.momentum is not None:
            # gradient-free (EMA teacher): statistics pass without a store, then convolution + BatchNorm + ReLU in one pass
            from .dwconv import dwconv3x3_bn_act_nhwc
            c = self.conv
            xh = x.permute(0, 2, 3, 1)
            y = dwconv3x3_bn_act_nhwc(xh if xh.is_contiguous() else xh.contiguous(), c.weight, c.bias, c.dilation[0], self.bn,
                                      act == 1)
            return y.permute(0, 3, 1, 2)
        if self._dw_stats_ok(x, cd):
            c = self.conv
            sums = torch.empty(2 * c.out_channels + 1, dtype=torch.float64, device=x.device)
            return bnk.bn_act_train(self._conv2d(x, c.weight, c.bias, stats=sums), self.bn, act, cd, out=out, sums=sums)
        return bnk.bn_act_train(self._conv_train(x, cd), self.bn, act, cd, out=out)

    def forward(self, x, out=None):
        if isinstance(x, (list, tuple)):
            # a channel concatenation handed over as its parts (the matcher's decoders: cat(correlation, flow, ...)): on the
            # gradient-free split-bf16 path the parts go straight into the convolution's operand (split32.conv2d_parts)
            c = getattr(self, "conv", None)
            if c is not None and out is None and x[0].is_cuda and c.groups == 1 and not torch.is_grad_enabled() \
                    and (not self.use_norm or not self.training) and self.act_slope in (0.0, LEAKY_SLOPE) \
                    and all(t.dtype == torch.float32 for t in x) and not torch.is_autocast_enabled("cuda") \
                    and c.stride[0] == c.stride[1] and c.padding[0] == c.padding[1] and c.dilation[0] == c.dilation[1]:
                from . import split32
                if split32.usable(*x):
                    w, b = self.folded() if self.use_norm else (c.weight, c.bias)
                    y = split32.conv2d_parts(list(x), w, b, c.stride[0], c.padding[0], c.dilation[0],
                                             act={None: 0, 'relu': 1, 'leaky': 3}[self.act])
                    if y is not None:
                        return y
            x = torch.cat(list(x), 1)
        if self.depthwise_separable:
            return self.pointwise_conv(self.depthwise_conv(x), out=out)
        c = self.conv
        if out is not None:
            from .params import compute_dtype
            return self._bn_train(x, compute_dtype(x), out=out)
        if x.is_cuda and c.groups == 1 and not torch.is_grad_enabled() and (not self.use_norm or not self.training) \
                and self.act_slope in (0.0, LEAKY_SLOPE):
            # gradient-free, BatchNorm in eval mode (the frozen matcher): ONE launch of the hand-written implicit-GEMM
            # kernel -- folded norm in the weights, bias and ReLU / LeakyReLU in its epilogue (refign_amd/conv.py)
            from .conv import conv2d_mfma
            from .params import compute_dtype
            w, b = self.folded() if self.use_norm else (c.weight, c.bias)
            y = conv2d_mfma(x, w, b, c.stride, c.padding, c.dilation, act=self.act, dtype=compute_dtype(x))
            if y is not None:
                return y
        if x.is_cuda and c.groups == 1 and not torch.is_grad_enabled() and (not self.use_norm or not self.training) \
                and self.act_slope in (0.0, LEAKY_SLOPE) and x.dtype == torch.float32 and not torch.is_autocast_enabled("cuda"):
            # the same block on fp32 tensors (parity mode; the matcher's head inside the timed step, align.HEAD_SPLIT): split-bf16
            # products, folded norm, bias and activation in the epilogue of the one launch
            from . import split32
            if split32.usable(x):
                w, b = self.folded() if self.use_norm else (c.weight, c.bias)
                y = split32.conv2d(x, w, b, c.stride, c.padding, c.dilation, act={None: 0, 'relu': 1, 'leaky': 3}[self.act])
                if y is not None:
                    return y
        if self.use_norm and not self.training and not torch.is_grad_enabled():
            x = self._conv2d(x, *self.folded())
        else:
            if x.is_cuda and self.use_norm and self.training and \
                    (self.act in (None, 'relu') or (self.act == 'leaky' and self.act_slope == LEAKY_SLOPE)) and \
                    os.environ.get("RFN_BN_KERNEL", "1") != "0":
                # decode heads (student and EMA teacher run BatchNorm with batch statistics): convolution on the
                # hand-written kernels where they exist for the pass, then ONE fused BatchNorm(train) + ReLU (csrc/bn.hip)
                from . import bn as bnk
                from .params import compute_dtype
                cd = compute_dtype(x)
                if bnk.usable(x, self.bn, cd, channels=c.out_channels, count=self._out_count(x)):
                    return self._bn_train(x, cd)
            x = self._conv2d(x, c.weight, c.bias)
            if self.use_norm:
                x = self.bn(x)
        if self.act == 'leaky':
            x = F.leaky_relu(x, self.act_slope, inplace=True)
        elif self.act == 'relu':
            x = F.relu(x, inplace=True)
        return x


class MLP(nn.Module):
    """Linear embedding of an NCHW map into tokens (models/modules.py:59-68): (B,C,H,W) -> (B,H*W,embed_dim)."""

    def __init__(self, input_dim=2048, embed_dim=768):
        super().__init__()
        from .linear import Linear
        self.proj = Linear(input_dim, embed_dim)

    def forward(self, x):
        return self.proj(x.flatten(2).transpose(1, 2))


class DropPath(nn.Module):
    """Stochastic depth per sample (models/modules.py:564-596)."""

    def __init__(self, drop_prob: float = 0., scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask

    def residual(self, x, y):
        """x + drop_path(y) in ONE elementwise kernel (addcmul) instead of a broadcast multiply and an add: the MiT
        residual stream is pure HBM traffic (2 x 52 blocks x 4 passes per step).  Same random draw as forward()."""
        if self.drop_prob == 0. or not self.training:
            return x + y
        keep = 1 - self.drop_prob
        mask = y.new_empty((y.shape[0],) + (1,) * (y.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        if x.dtype != y.dtype:
            return x + y * mask
        return torch.addcmul(x, y, mask)
