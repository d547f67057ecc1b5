"""Segmentation networks of the Refign step: MiT encoder, DAFormer / SegFormer decode heads, HRDA multi-resolution
fusion, pixel-weighted cross entropy.

Host mirror of (brdav/refign):
  models/backbones/mix_transformer.py   MixVisionTransformer(model_type, pretrained, ...)            (a18)
  models/heads/daformer.py              DAFormerHead(in_channels, in_index, num_classes, ...)       (a19)
  models/heads/segformer.py             SegFormerHead(...)  -- HRDA scale attention                 (a20)
  models/hrda.py                        hrda_backbone / hrda_head decorators and crop helpers       (a21)
  models/losses.py:10-22                PixelWeightedCrossEntropyLoss                               (a23)

Module trees and state_dict keys are the reference's (`block3.17.attn.sr.weight`, `mlp.dwconv.dwconv.bias`,
`fuse_layer.aspp_modules.2.depthwise_conv.bn.running_var`, `linear_fuse.conv.weight`, ...), so `mit_b5.pth` /
Lightning checkpoints load with strict=True.  Execution differs from the reference where it matters on MI355X:
tokens stay (B, N, C) with a single NCHW<->NHWC conversion per stage boundary, attention goes through one fused
scaled-dot-product call (long Q, <=2040 keys after spatial reduction, head_dim 64) instead of materialising the N x Nkv
score matrix, and K/V come from one fused projection.  Dense GEMM/conv/attention currently run on the ROCm libraries
behind torch (hipBLASLt / MIOpen / fused SDPA); see DESIGN.md for which of them are scheduled to become hand-written
MFMA kernels.
"""
import math
import os
import random
from functools import partial, wraps
from typing import Callable, List, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import f8 as _f8
from . import linear as _linear
from . import mfma
from ._tensor import const_tensor
from .align import BaseHead
from .conv import Conv2d, patch_conv_tokens
from .layernorm import LayerNorm
from .layers import MLP, ConvBNReLU, DropPath
from .linear import Linear
from .upcat import upsample_concat

# ---------------------------------------------------------------------------------------------------------------------
# MiT (SegFormer encoder)
# ---------------------------------------------------------------------------------------------------------------------
_MIT = {  # embed_dims, depths   (heads [1,2,5,8], mlp ratio 4, sr [8,4,2,1], qkv_bias, LN eps 1e-6 for all)
    'mit_b0': ([32, 64, 160, 256], [2, 2, 2, 2]),
    'mit_b1': ([64, 128, 320, 512], [2, 2, 2, 2]),
    'mit_b2': ([64, 128, 320, 512], [3, 4, 6, 3]),
    'mit_b3': ([64, 128, 320, 512], [3, 4, 18, 3]),
    'mit_b4': ([64, 128, 320, 512], [3, 8, 27, 3]),
    'mit_b5': ([64, 128, 320, 512], [3, 6, 40, 3]),
}
_MIT_HEADS = [1, 2, 5, 8]
_MIT_SR = [8, 4, 2, 1]


# ---------------------------------------------------------------------------------------------------------------------
# gradient-readiness marks (data-parallel all-reduce overlapped with the backward pass, refign_amd/trainer.py)
# ---------------------------------------------------------------------------------------------------------------------
_GRAD_READY_CB = None          # callable(tag) installed by the trainer for the LAST backward pass of a step


class _GradMark(torch.autograd.Function):
    """Identity whose backward tells the trainer that every parameter downstream of this point (later MiT stages, decode
    heads) has its final gradient: their slice of the flat gradient buffer can go on the wire while the backward of the
    earlier stages is still running."""

    @staticmethod
    def forward(ctx, x, tag):
        ctx.tag = tag
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        cb = _GRAD_READY_CB
        if cb is not None:
            cb(ctx.tag)
        return g, None


def grad_mark(x, tag):
    return _GradMark.apply(x, tag) if (torch.is_grad_enabled() and x.requires_grad) else x


class DWConv(nn.Module):
    """3x3 depthwise conv on tokens (mix_transformer.py:556-568); parameter path `dwconv.weight`."""

    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, x, H, W):
        B, N, C = x.shape
        if x.is_cuda and C % 8 == 0:
            # tokens ARE the channels-last map: hand-written HIP stencil, no NCHW round trip (csrc/dwconv.hip)
            from .dwconv import dwconv3x3_tokens
            return dwconv3x3_tokens(x, self.dwconv.weight, self.dwconv.bias, H, W)
        y = self.dwconv(x.transpose(1, 2).reshape(B, C, H, W))      # host-side formulation (CPU unit tests only)
        return y.flatten(2).transpose(1, 2)


class Mlp(nn.Module):
    """Mix-FFN (mix_transformer.py:79-103): fc1 -> depthwise 3x3 -> GELU (exact erf) -> fc2."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.act = act_layer()
        self.fc2 = Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x, H, W, res=None, rowscale=None):
        if x.is_cuda and not torch.is_grad_enabled() and x.dtype == torch.bfloat16 and type(self.act) is nn.GELU \
                and self.act.approximate == 'none' and self.drop.p == 0. and type(self.fc1) is Linear and not _f8.active():
            # gradient-free passes (EMA teacher's 40 views, ImageNet encoder): fc1 + depthwise 3x3 + GELU in ONE kernel, the 4C-wide
            # pre-activation never reaches HBM (csrc/mixffn.hip); fc2 (+ residual) follows as before
            from .dwconv import ffn_fc1_dw_gelu
            a = ffn_fc1_dw_gelu(x, self.fc1, self.dwconv.dwconv, H, W)
            if a is not None:
                return self.fc2(a, res=res, rowscale=rowscale) if res is not None else self.fc2(a)
        x = self.fc1(x)
        if x.is_cuda and x.shape[-1] % 8 == 0 and type(self.act) is nn.GELU and self.act.approximate == 'none':
            from .dwconv import dwconv3x3_gelu_tokens           # depthwise conv + GELU in one pass (csrc/dwconv.hip)
            dw = self.dwconv.dwconv
            if torch.is_grad_enabled() and x.requires_grad and self.drop.p == 0. and x.dtype in (torch.bfloat16, torch.float16) \
                    and os.environ.get("RFN_FUSED_GELU_BWD", "1") == "1":
                # the activation + its pre-activation; fc2's input-gradient GEMM applies gelu' in its epilogue instead of a
                # gelu_backward pass over the 4C-wide hidden tensor (test_mix_ffn_gelu_backward_in_the_fc2_dgrad_epilogue).
                # Neutral on the step in rounds 2-3 (182.4 / 183.0 vs 182.4 / 182.5 ms); on the round-4 kernels -0.5 ms alone and
                # -1.5 ms together with 1 024 BatchNorm workgroups and the 2 000-tile GEMM threshold, both constants now (three alternating runs each,
                # profiles/r04_knob_ab.txt): on since the end of round 4.  RFN_FUSED_GELU_BWD=0: the separate pass.
                a, z = dwconv3x3_gelu_tokens(x, dw.weight, dw.bias, H, W, with_z=True)
                if res is not None:
                    return self.fc2(a, res=res, rowscale=rowscale, z=z)
                return self.fc2(a, z=z)
            x = self.drop(dwconv3x3_gelu_tokens(x, dw.weight, dw.bias, H, W))
        else:
            x = self.drop(self.act(self.dwconv(x, H, W)))
        if res is not None and self.drop.p == 0.:
            return self.fc2(x, res=res, rowscale=rowscale)       # residual (+ drop-path scale) in the GEMM epilogue
        y = self.drop(self.fc2(x))
        return y if res is None else _residual(res, y, rowscale)


_SDPA_BACKEND = None          # (attention never goes to the fused-SDPA library on 16-bit or fp32 HIP tensors; the branch below is
#                               what CPU tensors take -- checkpoint / config tests on the host)
_FUSED_UPCAT = True       # decode heads: up-sampling + concat in one kernel


def _fused_upcat_here():
    """gradient-free passes always; under autograd with the gather backward kernel (_FUSED_UPCAT = False: the unfused
    graph -- the library's bilinear backward on channel slices of the fused gradient was 34 ms/step slower than that)"""
    return _FUSED_UPCAT
_SR_AS_LINEAR = True     # spatial-reduction conv as a Linear over patches


class Attention(nn.Module):
    """Efficient self-attention with spatial-reduction K/V (mix_transformer.py:106-164)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., sr_ratio=1):
        super().__init__()
        assert dim % num_heads == 0, f'dim {dim} should be divided by num_heads {num_heads}.'
        self.dim, self.num_heads = dim, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.q = Linear(dim, dim, bias=qkv_bias)
        self.kv = Linear(dim, dim * 2, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.sr_ratio = sr_ratio
        if sr_ratio > 1:
            self.sr = Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = LayerNorm(dim)

    def forward(self, x, H, W, res=None, rowscale=None, x_kv=None):
        """`x_kv`: a second handle on the same tensor for the key / value path (_norm_pass(..., fan=2): the two gradients of the
        block's LayerNorm output are then summed inside its backward kernel, not by an element-wise launch of the engine)."""
        B, N, C = x.shape
        h, d = self.num_heads, C // self.num_heads
        q = self.q(x)                                                        # (B,N,C) = (B,N,h,d)
        if x_kv is not None:
            x = x_kv
        if self.sr_ratio > 1:
            r = patch_conv_tokens(x, H, W, self.sr) if _SR_AS_LINEAR else None   # (B, N/sr^2, C) tokens directly
            if r is None:
                r = self.sr(x.transpose(1, 2).reshape(B, C, H, W)).flatten(2).transpose(1, 2)
            x = self.norm(r)
        kv = self.kv(x)                                                      # (B,Nkv,2C) = (B,Nkv,2,h,d)
        p = self.attn_drop.p if self.training else 0.0
        # hand-written MFMA attention (csrc/attn.hip): head_dim 64, 16-bit operands, no attention dropout
        o = mfma.attention(q, kv, h, self.scale) if (d == 64 and p == 0.0 and _SDPA_BACKEND is None) else None
        if o is None and q.is_cuda and q.dtype == torch.float32 and p == 0.0 and _SDPA_BACKEND is None:
            from . import split32
            if split32.usable(q, kv) and d in (32, 64):    # fp32 parity mode: the fp32 matrix-pipe attention kernel (csrc/attn32.hip)
                o = split32.attention(q, kv, h, self.scale)
        if o is None:
            mfma.note_library("sdpa", q, kv)
            q = q.view(B, N, h, d).transpose(1, 2)                           # (B,h,N,d)
            # unbind, not kv[0] / kv[1]: its backward is ONE stack of (dK, dV) instead of two zero-fills, two slice
            # copies and an add
            k, v = kv.view(B, -1, 2, h, d).permute(2, 0, 3, 1, 4).unbind(0)  # (B,h,Nkv,d) each
            if _SDPA_BACKEND is None:
                o = F.scaled_dot_product_attention(q, k, v, dropout_p=p, scale=self.scale)
            else:                                      # (module constant: a torch.nn.attention.SDPBackend)
                with torch.nn.attention.sdpa_kernel([_SDPA_BACKEND]):
                    o = F.scaled_dot_product_attention(q, k, v, dropout_p=p, scale=self.scale)
            o = o.transpose(1, 2).reshape(B, N, C)
        if res is not None and self.proj_drop.p == 0.:
            return self.proj(o, res=res, rowscale=rowscale)
        y = self.proj_drop(self.proj(o))
        return y if res is None else _residual(res, y, rowscale)


def _residual(res, y, rowscale):
    if rowscale is None:
        return res + y
    return torch.addcmul(res, y, rowscale.to(y.dtype).view((-1,) + (1,) * (y.dim() - 1)))


_LN_PASS = True


_LN_FAN2 = True


def _norm_pass(norm, x, fan=1):
    """(norm(x), x') where x' is x to be used as the residual operand of the branch: under autograd on the GPU the two
    gradients of x (through the LayerNorm and through the residual add) are then summed inside the LayerNorm-backward kernel
    (layernorm.layer_norm_pass) instead of by an element-wise launch of the autograd engine.
    fan=2: ((n, n'), x') -- two handles on norm(x) for the two consumers inside the attention module (q projection, key / value
    path), whose gradients meet in the same kernel (layernorm.layer_norm_pass2)."""
    if _LN_PASS and x.is_cuda and torch.is_grad_enabled() and x.requires_grad and type(norm) is LayerNorm and x.shape[-1] % 8 == 0 \
            and x.shape[-1] <= 1024 and x.dtype in (torch.float32, torch.bfloat16) and len(norm.normalized_shape) == 1:
        from .layernorm import layer_norm_pass, layer_norm_pass2
        if fan == 2 and _LN_FAN2:
            n, nb, xa = layer_norm_pass2(x, norm.weight, norm.bias, norm.eps)
            return (n, nb), xa
        n, xa = layer_norm_pass(x, norm.weight, norm.bias, norm.eps)
        return ((n, None), xa) if fan == 2 else (n, xa)
    n = norm(x)
    return ((n, None), x) if fan == 2 else (n, x)


class Block(nn.Module):
    """Pre-norm transformer block with stochastic depth (mix_transformer.py:167-207)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=LayerNorm, sr_ratio=1):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop, sr_ratio=sr_ratio)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x, H, W, masks=None, masks32=None):
        if x.is_cuda and not torch.is_grad_enabled() and (masks is None) == (masks32 is None) and \
                (masks is not None or not (self.training and isinstance(self.drop_path, DropPath)
                                           and self.drop_path.drop_prob > 0.)):
            # gradient-free passes (EMA teacher, ImageNet features, inference): the residual add and the per-sample
            # stochastic-depth scale ride in the epilogue of the proj / fc2 GEMMs
            if _f8.active() and _f8.block_supported(self, x):
                return _f8.block_forward(self, x, H, W, masks32)      # K5: the block on the fp8 matrix-core kernels
            x = self.attn(self.norm1(x), H, W, res=x, rowscale=None if masks32 is None else masks32[0])
            return self.mlp(self.norm2(x), H, W, res=x, rowscale=None if masks32 is None else masks32[1])
        # (training: the block's weight gradients are queued during the backward pass and launched as one group when it has left
        # the block -- mfma.deferred_wgrads / rfn_gemm_tn_grouped)
        x = mfma.wgrad_mark(x)
        if masks is not None:                       # pre-drawn stochastic-depth masks (MixVisionTransformer)
            if x.is_cuda and masks32 is not None and _linear._FUSED_RESIDUAL:
                # training (RFN_FUSED_RESIDUAL, on): the same fusion under autograd (linear._LinearFn: residual + per-sample scale in the proj /
                # fc2 GEMM epilogue; in the backward the scale rides in the input- and weight-gradient kernels)
                (n, nb), xa = _norm_pass(self.norm1, x, fan=2)
                x = self.attn(n, H, W, res=xa, rowscale=masks32[0], x_kv=nb)
                n, xa = _norm_pass(self.norm2, x)
                return self.mlp(n, H, W, res=xa, rowscale=masks32[1])
            x = torch.addcmul(x, self.attn(self.norm1(x), H, W), masks[0])
            return torch.addcmul(x, self.mlp(self.norm2(x), H, W), masks[1])
        dp = self.drop_path
        if x.is_cuda and _linear._FUSED_RESIDUAL and not (self.training and isinstance(dp, DropPath) and dp.drop_prob > 0.):
            (n, nb), xa = _norm_pass(self.norm1, x, fan=2)
            x = self.attn(n, H, W, res=xa, x_kv=nb)
            n, xa = _norm_pass(self.norm2, x)
            return self.mlp(n, H, W, res=xa)
        res = dp.residual if isinstance(dp, DropPath) else torch.add
        x = res(x, self.attn(self.norm1(x), H, W))
        return res(x, self.mlp(self.norm2(x), H, W))


class OverlapPatchEmbed(nn.Module):
    """Overlapping patch embedding: strided conv (k=7/s=4 or k=3/s=2, pad k//2) + LayerNorm (default eps 1e-5)
    (mix_transformer.py:210-242)."""

    def __init__(self, img_size=224, patch_size=7, stride=4, in_chans=3, embed_dim=768):
        super().__init__()
        self.proj = Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.norm = LayerNorm(embed_dim)

    def forward(self, x):
        x = self.proj(x)
        H, W = x.shape[-2:]
        return self.norm(x.flatten(2).transpose(1, 2)), H, W


class MixVisionTransformer(nn.Module):
    """mix_transformer.py:245-553.  forward(x) -> [C1@1/4, C2@1/8, C3@1/16, C4@1/32] NCHW maps."""

    def __init__(self, model_type: str, pretrained: Optional[str] = None, img_size: int = 224, in_chans: int = 3,
                 qk_scale: Optional[float] = None, drop_rate: float = 0., attn_drop_rate: float = 0.,
                 drop_path_rate: float = 0.1, freeze_patch_embed: bool = False):
        super().__init__()
        dims, depths = _MIT[model_type]
        self.model_type, self.depths = model_type, depths
        norm_layer = partial(LayerNorm, eps=1e-6)
        dpr = torch.linspace(0, drop_path_rate, sum(depths)).tolist()        # stochastic depth decay rule
        cur, cin = 0, in_chans
        for s in range(4):
            setattr(self, f"patch_embed{s + 1}",
                    OverlapPatchEmbed(img_size // (1 if s == 0 else 2 ** (s + 1)), 7 if s == 0 else 3,
                                      4 if s == 0 else 2, cin, dims[s]))
            setattr(self, f"block{s + 1}", nn.ModuleList([
                Block(dims[s], _MIT_HEADS[s], 4, True, qk_scale, drop_rate, attn_drop_rate, dpr[cur + i],
                      norm_layer=norm_layer, sr_ratio=_MIT_SR[s]) for i in range(depths[s])]))
            setattr(self, f"norm{s + 1}", norm_layer(dims[s]))
            cur += depths[s]
            cin = dims[s]
        if freeze_patch_embed:
            self.patch_embed1.requires_grad = False                          # (sic) mix_transformer.py:494-495
        self.init_weights(pretrained)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Conv2d):
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
            m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                m.bias.data.zero_()

    def init_weights(self, pretrained=None):
        """mix_transformer.py:445-479: local file (or $TORCH_HOME/hub/<path>); strips `backbone.`, drops `head.*`."""
        if pretrained is None:
            self.apply(self._init_weights)
            return
        names = {'imagenet': f'{self.model_type}.pth', 'cityscapes': f'{self.model_type}.pth'}
        path = names.get(pretrained, pretrained)
        for cand in (path, os.path.join(os.environ.get('TORCH_HOME', ''), 'hub', path),
                     os.path.join('pretrained_models', path)):
            if os.path.exists(cand):
                ckpt = torch.load(cand, map_location='cpu')
                break
        else:
            raise FileNotFoundError(f"MiT weights '{pretrained}' not found locally (no network access)")
        sd = ckpt.get('state_dict', ckpt.get('model', ckpt))
        if any(k.startswith('backbone.') for k in sd):
            sd = {k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}
        self.load_state_dict({k: v for k, v in sd.items() if not k.startswith('head.')}, strict=True)

    def reset_drop_path(self, drop_path_rate):
        dpr = torch.linspace(0, drop_path_rate, sum(self.depths)).tolist()
        cur = 0
        for s in range(4):
            for i, blk in enumerate(getattr(self, f"block{s + 1}")):
                blk.drop_path.drop_prob = dpr[cur + i]
            cur += self.depths[s]

    def _drop_path_masks(self, x):
        """All stochastic-depth masks of one forward pass in three launches (Bernoulli, scale, cast) instead of two
        tiny kernels per residual branch (2 x 52 branches x 4 passes per step on MiT-B5).  Per sample and per branch,
        scaled by 1/keep, exactly what DropPath.forward draws (models/modules.py:564-596); only the order in which the
        generator is consumed differs.  (n_branches, B, 1, 1) in the compute dtype, or None when nothing is dropped."""
        blocks = [b for s in range(1, 5) for b in getattr(self, f"block{s}")]
        keep = [1.0 - (b.drop_path.drop_prob if (isinstance(b.drop_path, DropPath) and b.training) else 0.0)
                for b in blocks for _ in range(2)]
        if not x.is_cuda or all(k == 1.0 for k in keep) or any(k <= 0.0 for k in keep) or \
                not all(getattr(b.drop_path, "scale_by_keep", True) for b in blocks):
            return None
        k = const_tensor(keep, x, dtype=torch.float32).view(-1, 1)
        m = (torch.bernoulli(k.expand(-1, x.shape[0])) / k).contiguous()
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        return m.to(dt).view(len(keep), x.shape[0], 1, 1), m              # compute-dtype view + the fp32 (n, B) masks

    def forward_features(self, x):
        B, outs = x.shape[0], []
        mm, i = self._drop_path_masks(x), 0
        masks, masks32 = (None, None) if mm is None else mm
        for s in range(1, 5):
            if s > 1:
                x = grad_mark(x, f"stage{s}")        # backward passing here: stages s.. and everything after are done
            x, H, W = getattr(self, f"patch_embed{s}")(x)
            for blk in getattr(self, f"block{s}"):
                x = blk(x, H, W, None if masks is None else masks[i:i + 2], None if masks32 is None else masks32[i:i + 2])
                i += 2
            x = getattr(self, f"norm{s}")(x)
            # tokens (B, H*W, C) ARE the channels-last image of the (B, C, H, W) map: hand it on as that view on the GPU --
            # the next patch embedding and the decode heads' token embeddings consume channels-last memory in place
            # (the reference's reshape + permute + contiguous is two full copies per stage and pass)
            x = x.view(B, H, W, -1).permute(0, 3, 1, 2)
            if not x.is_cuda:
                x = x.contiguous()
            outs.append(x)
        return outs

    def forward(self, x):
        return self.forward_features(x)


# ---------------------------------------------------------------------------------------------------------------------
# decode heads
# ---------------------------------------------------------------------------------------------------------------------
def _mmseg_init(module):
    for m in module.modules():
        if isinstance(m, ConvBNReLU) and not m.depthwise_separable:
            nn.init.kaiming_normal_(m.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
            if m.conv.bias is not None:
                nn.init.zeros_(m.conv.bias)
            if m.use_norm and m.bn.weight is not None:
                nn.init.ones_(m.bn.weight)
                nn.init.zeros_(m.bn.bias)


def _up(x, size):
    return F.interpolate(x, size=size, mode='bilinear', align_corners=False)


def _up_logits(x, size):
    """_up for class logits that go straight into the loss: re-laid out to NCHW at the LOW resolution, so that the
    up-sampled tensor is NCHW-contiguous and log_softmax does not copy it (uda._upsample_logits)."""
    if x.is_cuda:
        x = x.contiguous()
    return _up(x, size)


def _class_logits(conv, y):
    """The 19-class 1x1 convolution of a decode head (daformer.py:225, segformer.py:109).  On the GPU in a 16-bit pass it
    runs on the implicit-GEMM kernels (output channels padded to 64 inside, conv._ConvMfmaFn / conv2d_mfma); elsewhere
    it is the module itself."""
    if y.is_cuda and conv.kernel_size == (1, 1):
        from .conv import conv2d_mfma, conv2d_mfma_grad
        from .params import compute_dtype
        cd = compute_dtype(y)
        if cd in (torch.float16, torch.bfloat16):
            if torch.is_grad_enabled() and (conv.weight.requires_grad or y.requires_grad):
                o = conv2d_mfma_grad(y, conv.weight, conv.bias, 1, 0, 1, cd)
            else:
                o = conv2d_mfma(y, conv.weight, conv.bias, 1, 0, 1, dtype=cd)
            if o is not None:
                return o
        if cd == torch.float32:
            from . import split32
            if split32.usable(y):
                o = split32.conv2d(y.float(), conv.weight, conv.bias, 1, 0, 1)
                if o is not None:
                    return o
        mfma.note_library("conv2d.autograd" if torch.is_grad_enabled() else "conv2d", y, conv.weight)
    return conv(y)


class DepthwiseSeparableASPPModule(nn.ModuleList):
    """daformer.py:10-62: branch 0 is a 1x1 ConvBNReLU, dilated branches are depthwise-separable 3x3."""

    def __init__(self, dilations, in_channels, channels, norm_layer, activation_layer):
        super().__init__()
        self.dilations = dilations
        for d in dilations:
            if d == 1:
                self.append(ConvBNReLU(in_channels, channels, 1, dilation=1, padding=0, norm_layer=norm_layer,
                                       activation_layer=activation_layer))
            else:
                self.append(ConvBNReLU(in_channels, channels, 3, dilation=d, padding=d, norm_layer=norm_layer,
                                       activation_layer=activation_layer, depthwise_separable=True))

    def forward(self, x):
        return [m(x) for m in self]


_ASPP_NOCAT = True
# the three dilated branches from ONE LDS-resident copy of the input (csrc/dwconv.hip: dwconv3x3_tri_kernel): correct, every
# input byte fetched once at HBM speed (0.53 ms for the teacher's 2.65 GB), but the passes are VALU-bound (bf16 -> fp32
# conversions + packed FMAs + zero-padding selects: 2.9 ms of 3.3), so it measures 8.6 ms against 8.1 ms for six single-branch
# passes (profiles/r04_aspp_try.txt): kept, tested, OFF by default
_ASPP_TRI = False


class ASPPWrapper(nn.Module):
    """daformer.py:65-126 with sep=True, pool=False, no context layer (the DAFormer configuration)."""

    def __init__(self, in_channels, channels, sep, dilations, pool, norm_layer, activation_layer, context_cfg=None):
        super().__init__()
        if not sep or pool or context_cfg is not None:
            raise NotImplementedError("DAFormer uses sep=True, pool=False, context_cfg=None (daformer.py:181-182)")
        self.dilations = dilations
        self.image_pool = None
        self.context_layer = None
        self.aspp_modules = DepthwiseSeparableASPPModule(dilations, in_channels, channels, norm_layer,
                                                         activation_layer)
        self.bottleneck = ConvBNReLU(len(dilations) * channels, channels, kernel_size=3, padding=1,
                                     norm_layer=norm_layer, activation_layer=activation_layer)

    def forward(self, x):
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)   # depthwise branches + 1x1 convs run channels-last
        mods = list(self.aspp_modules)
        if x.is_cuda and not torch.is_grad_enabled() and _ASPP_NOCAT and all(m.fused_out_ok(x) for m in mods):
            # gradient-free (the EMA teacher: 42 maps): every branch's BatchNorm + ReLU writes its 256 channels straight
            # into the concatenated channels-last tensor -- the torch.cat was 5.3 GB of traffic, 1.1 ms per step
            from .params import compute_dtype
            B, _, H, W = x.shape
            ch = [(m.pointwise_conv if m.depthwise_separable else m).conv.out_channels for m in mods]
            if all(c % 8 == 0 for c in ch):
                cat = torch.empty((B, H, W, sum(ch)), dtype=compute_dtype(x), device=x.device)
                # round 4: the three dilated depthwise branches (6 / 12 / 18 = g, 2 g, 3 g) of the same input from ONE statistics
                # pass and ONE convolution + BatchNorm + ReLU pass over it (csrc/dwconv.hip dwconv3x3_tri_kernel) instead of
                # three of each
                dw_out = {}
                sep = [m for m in mods if m.depthwise_separable]
                if _ASPP_TRI and len(sep) == 3 and compute_dtype(x) == torch.bfloat16:
                    from .dwconv import dwconv3x3_bn_act_nhwc_tri, tri_usable
                    dws = [m.depthwise_conv for m in sep]
                    xh = x.permute(0, 2, 3, 1)
                    if xh.dtype == torch.bfloat16 and xh.is_contiguous() and all(d.use_norm and d.training and d.act in (None, 'relu')
                                                                                 and d.act == dws[0].act for d in dws) \
                            and tri_usable(xh, [d.conv for d in dws], [d.bn for d in dws]):
                        ys = dwconv3x3_bn_act_nhwc_tri(xh, [d.conv for d in dws], [d.bn for d in dws], dws[0].act == 'relu')
                        dw_out = {id(m): y.permute(0, 3, 1, 2) for m, y in zip(sep, ys)}
                o = 0
                for m, c in zip(mods, ch):
                    out = cat[..., o:o + c].permute(0, 3, 1, 2)
                    if id(m) in dw_out:
                        m.pointwise_conv(dw_out[id(m)], out=out)
                    else:
                        m(x, out=out)
                    o += c
                return self.bottleneck(cat.permute(0, 3, 1, 2))
        return self.bottleneck(torch.cat(self.aspp_modules(x), dim=1))


class DAFormerHead(BaseHead):
    """daformer.py:152-227: per-stage Linear embed -> bilinear to 1/4 -> concat -> sep-ASPP -> Dropout2d -> 1x1."""

    def __init__(self, in_channels: List[int], in_index: Union[List[int], int], num_classes: int,
                 input_transform: Optional[str] = None, channels: int = 256, dropout_ratio: float = 0.1,
                 embed_dims: int = 256):
        super().__init__(num_classes, in_index, input_transform)
        self.in_channels, self.channels = in_channels, channels
        if isinstance(embed_dims, int):
            embed_dims = [embed_dims] * len(in_channels)
        self.embed_layers = nn.ModuleDict({str(i): MLP(input_dim=c, embed_dim=e)
                                           for i, (c, e) in enumerate(zip(in_channels, embed_dims))})
        self.fuse_layer = ASPPWrapper(sum(embed_dims), channels, sep=True, dilations=(1, 6, 12, 18), pool=False,
                                      norm_layer=nn.BatchNorm2d, activation_layer=nn.ReLU)
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        self.conv_seg = nn.Conv2d(channels, num_classes, kernel_size=1)
        nn.init.normal_(self.conv_seg.weight, mean=0, std=0.01)
        nn.init.zeros_(self.conv_seg.bias)
        _mmseg_init(self)

    def forward(self, x):
        x = self._transform_inputs(x)
        size = x[0].shape[2:]
        toks = [self.embed_layers[str(i)](f) for i, f in enumerate(x)]          # (n, h_l*w_l, embed) token maps
        # one pass (csrc/upcat.hip), forward and backward
        cat = upsample_concat(toks, [f.shape[2:] for f in x], size) if _fused_upcat_here() else None
        if cat is None:
            cs = []
            for c, f in zip(toks, x):
                n, _, h, w = f.shape
                c = c.transpose(1, 2).reshape(n, -1, h, w)
                cs.append(c if (h, w) == tuple(size) else _up(c, size).to(c.dtype))
            cat = torch.cat(cs, dim=1)
        y = self.fuse_layer(cat)
        if self.dropout is not None:
            y = self.dropout(y)
        return _class_logits(self.conv_seg, y)


class SegFormerHead(BaseHead):
    """segformer.py:15-111 (all-MLP decoder; used as HRDA's scale-attention head)."""

    os: int = 4

    def __init__(self, in_channels: List[int], in_index: Union[List[int], int], num_classes: int,
                 input_transform: Optional[str] = None, channels: int = 256, dropout_ratio: float = 0.1):
        super().__init__(num_classes, in_index, input_transform)
        self.in_channels = in_channels
        c1, c2, c3, c4 = in_channels
        self.linear_c4 = MLP(input_dim=c4, embed_dim=channels)
        self.linear_c3 = MLP(input_dim=c3, embed_dim=channels)
        self.linear_c2 = MLP(input_dim=c2, embed_dim=channels)
        self.linear_c1 = MLP(input_dim=c1, embed_dim=channels)
        self.linear_fuse = ConvBNReLU(channels * 4, channels, kernel_size=1, norm_layer=nn.BatchNorm2d)
        self.linear_pred = nn.Conv2d(channels, num_classes, kernel_size=1)
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        nn.init.normal_(self.linear_pred.weight, mean=0, std=0.01)
        nn.init.zeros_(self.linear_pred.bias)
        _mmseg_init(self)

    def forward(self, inputs):
        c1, c2, c3, c4 = inputs                      # NB: takes the raw 4-tuple, no _transform_inputs (segformer.py:80)
        size = c1.shape[2:]

        feats = [c4, c3, c2, c1]                     # concat order of the reference (segformer.py:97)
        toks = [layer(f) for layer, f in zip((self.linear_c4, self.linear_c3, self.linear_c2, self.linear_c1), feats)]
        cat = upsample_concat(toks, [f.shape[2:] for f in feats], size) if _fused_upcat_here() else None
        if cat is None:
            parts = []
            for t, f in zip(toks, feats):
                n, _, h, w = f.shape
                t = t.transpose(1, 2).reshape(n, -1, h, w)
                parts.append(t if (h, w) == tuple(size) else _up(t, size))
            cat = torch.cat(parts, dim=1)
        y = self.linear_fuse(cat)
        if self.dropout is not None:
            y = self.dropout(y)
        return _class_logits(self.linear_pred, y)


# ---------------------------------------------------------------------------------------------------------------------
# HRDA multi-resolution wrappers (models/hrda.py)
# ---------------------------------------------------------------------------------------------------------------------
_PREDRAWN_CROPS = []
_DEVICE_CROPS = []          # crop offsets as DEVICE data (hipGraph replay of the student passes), see DeviceBox


class DeviceBox:
    """A crop box whose offset lives in device memory: `off` = int64 tensor (oy, ox), (h, w) python ints.
    extract_crop / hr_crop_slice / crop() of the reference bake the random HRDA crop offsets (hrda.py:22-27) into
    slices, i.e. into kernel arguments -- a captured graph would replay one fixed crop for ever.  With the offsets as
    device data the same operations are gathers / comparisons against tensors the host refreshes before each replay.
    Offsets are multiples of `divisible` (2 * head_os), so the reference's int(v / scale) box scaling
    (hrda.py:50-64) is the exact integer division used here."""

    def __init__(self, off, h, w, divisible):
        self.off, self.h, self.w, self.divisible = off, int(h), int(w), int(divisible)

    def _idx(self, n, which, scale=1):
        return torch.arange(n, device=self.off.device) + self.off[which] // scale

    def crop(self, x):
        """x[..., y1:y2, x1:x2]"""
        return x.index_select(-2, self._idx(self.h, 0)).index_select(-1, self._idx(self.w, 1))

    def mask(self, H, W, scale, like):
        """ones on the box scaled by 1/scale (hr_crop_slice), zeros elsewhere: (1, 1, H, W)"""
        assert self.divisible % int(scale) == 0
        hs, ws = int(self.h / scale), int(self.w / scale)
        ry = torch.arange(H, device=self.off.device) - self.off[0] // int(scale)
        rx = torch.arange(W, device=self.off.device) - self.off[1] // int(scale)
        m = ((ry >= 0) & (ry < hs))[:, None] & ((rx >= 0) & (rx < ws))[None, :]
        return m.to(like.dtype).view(1, 1, H, W)

    def insert(self, patch, size, scale):
        """zeros(size) with `patch` written at the box scaled by 1/scale (gather from the zero-padded patch)"""
        assert self.divisible % int(scale) == 0
        H, W = size
        hs, ws = patch.shape[-2:]
        ry = torch.arange(H, device=self.off.device) - self.off[0] // int(scale)
        rx = torch.arange(W, device=self.off.device) - self.off[1] // int(scale)
        ry = torch.where((ry >= 0) & (ry < hs), ry, torch.full_like(ry, hs))
        rx = torch.where((rx >= 0) & (rx < ws), rx, torch.full_like(rx, ws))
        return F.pad(patch, (0, 1, 0, 1)).index_select(-2, ry).index_select(-1, rx)


def push_device_crop(off, divisible):
    """The NEXT extract_crop takes its offsets from the device tensor `off` (int64, (oy, ox))."""
    _DEVICE_CROPS.append((off, divisible))


def draw_crop_offsets(H, W, crop_size, divisible=1):
    """The two `random.randrange` draws of extract_crop (hrda.py:22-27), or None when no draw happens (crop == image)."""
    if H == crop_size[-2] and W == crop_size[-1]:
        return None
    oy = random.randrange(0, int((max(H - crop_size[-2], 0) + 1) // divisible)) * divisible
    ox = random.randrange(0, int((max(W - crop_size[-1], 0) + 1) // divisible)) * divisible
    return int(oy), int(ox)


def predraw_crop(H, W, crop_size, divisible=1):
    """Draw the NEXT extract_crop's offsets now, keeping the position of those draws in the python `random` stream.
    The training step needs its third draw (the adapt_to_ref coin, segmentation_model.py:195) before the source forward
    that makes the first two, so that the teacher branch can start on a side stream; extract_crop then uses these."""
    _PREDRAWN_CROPS.append(((H, W, tuple(crop_size), divisible), draw_crop_offsets(H, W, crop_size, divisible)))


def extract_crop(img, crop_size, divisible=1):
    """Random crop with offsets that are multiples of `divisible` (hrda.py:9-34).  Uses python `random` like the
    reference, so the same seed gives the same box."""
    H, W = img.shape[-2:]
    assert crop_size[0] > 0 and crop_size[1] > 0
    if _DEVICE_CROPS:
        off, div = _DEVICE_CROPS.pop(0)
        assert div == divisible, (div, divisible)
        box = DeviceBox(off, crop_size[0], crop_size[1], divisible)
        return box.crop(img), [box]
    if _PREDRAWN_CROPS:
        key, off = _PREDRAWN_CROPS.pop(0)
        assert key == (H, W, tuple(crop_size), divisible), (key, (H, W, tuple(crop_size), divisible))
    else:
        off = draw_crop_offsets(H, W, crop_size, divisible)
    if off is None:
        return (0, H, 0, W)                                                   # (sic) hrda.py:20-21
    oy, ox = off
    y1, y2, x1, x2 = int(oy), int(oy + crop_size[0]), int(ox), int(ox + crop_size[1])
    return img[:, :, y1:y2, x1:x2], [[y1, y2, x1, x2]]


def scale_box(box, scale):
    """hrda.py:50-64 (int() truncation of each coordinate)."""
    return tuple(int(v / scale) for v in box)


def hr_crop_slice(crop_box, scale):
    y1, y2, x1, x2 = scale_box(crop_box, scale)
    return slice(y1, y2), slice(x1, x2)


def extract_slide_crop(img, crop_size):
    """Sliding crops with stride = crop/2, last crops clamped to the border (hrda.py:67-94)."""
    hc, wc = crop_size
    hs, ws = hc // 2, wc // 2
    H, W = img.shape[-2:]
    crops, boxes = [], []
    for iy in range(max(H - hc + hs - 1, 0) // hs + 1):
        for ix in range(max(W - wc + ws - 1, 0) // ws + 1):
            y2, x2 = min(iy * hs + hc, H), min(ix * ws + wc, W)
            y1, x1 = max(y2 - hc, 0), max(x2 - wc, 0)
            crops.append(img[:, :, y1:y2, x1:x2])
            boxes.append([y1, y2, x1, x2])
    return torch.cat(crops, dim=0), boxes


def hrda_backbone(self, head_os: int, is_teacher: bool = False) -> Callable:
    """Decorator for backbone.forward (hrda.py:97-136): low-res (x0.5 bilinear) view + HR crop(s) in ONE backbone
    pass; student in training = one random crop (offsets multiple of 2*head_os), teacher / eval = sliding crops."""
    def deco(fn: Callable) -> Callable:
        @wraps(fn)
        def inner(x, *args, **kwargs):
            lr = F.interpolate(x, scale_factor=0.5, mode='bilinear', align_corners=False)
            size = lr.shape[-2:]
            if self.training and not is_teacher:
                hr, boxes = extract_crop(x, size, head_os * 2.0)
            else:
                hr, boxes = extract_slide_crop(x, size)
            nl, nh = lr.shape[0], hr.shape[0]
            feats = fn(torch.cat((lr, hr)), *args, **kwargs)
            lr_feats, hr_feats = zip(*(torch.split(f, [nl, nh]) for f in feats))
            return lr_feats, hr_feats, boxes
        return inner
    return deco


_REJOIN = True


def _rejoin(a, b):
    """torch.cat((a, b)) for the two halves torch.split made of ONE tensor along dim 0 (hrda_backbone splits the backbone's
    batch into LR views and HR crops, hrda_head runs the decode head on both again, hrda.py:139-150): when they are still
    adjacent views of one dense buffer and nothing is to be differentiated, the joined batch is a view of that buffer
    -- no copy (the teacher's stage-1 features are 174 MB per step).  Anything else: torch.cat."""
    if _REJOIN and (not torch.is_grad_enabled() or not (a.requires_grad or b.requires_grad)) and a.dim() >= 2 \
            and a.dtype == b.dtype \
            and a.shape[1:] == b.shape[1:] and a.stride() == b.stride() and a.device == b.device \
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() \
            and b.storage_offset() == a.storage_offset() + a.shape[0] * a.stride(0):
        dense = sorted(zip(a.stride()[1:], a.shape[1:]))        # inner dims tile the sample without gaps, in any order
        run = 1
        for st, n in dense:
            if n != 1 and st != run:
                return torch.cat((a, b))
            run *= n
        if a.stride(0) == run:
            return a.as_strided((a.shape[0] + b.shape[0], *a.shape[1:]), a.stride(), a.storage_offset())
    return torch.cat((a, b))


def hrda_head(self, hrda_scale_attention: nn.Module, head_os: int, is_teacher: bool = False) -> Callable:
    """Decorator for head.forward (hrda.py:139-235): scale attention from the LR features, fusion of LR logits and
    HR crop logits.  Student/train returns (logits, hr_logits, crop_box); teacher/eval returns logits."""
    def deco(fn: Callable) -> Callable:
        @wraps(fn)
        def inner(inp, *args, **kwargs):
            lr_feats, hr_feats, boxes = inp
            att = torch.sigmoid(hrda_scale_attention(lr_feats))
            nl, nh = lr_feats[0].shape[0], hr_feats[0].shape[0]
            seg = fn([_rejoin(*p) for p in zip(lr_feats, hr_feats)], *args, **kwargs)
            lr_seg, hr_seg = torch.split(seg, [nl, nh])
            if self.training and not is_teacher and isinstance(boxes[0], DeviceBox):
                box = boxes[0]
                att = att * box.mask(lr_seg.shape[2], lr_seg.shape[3], 2.0 * head_os, lr_seg)
                up_lr = F.interpolate((1 - att) * lr_seg, scale_factor=2, mode='bilinear', align_corners=False)
                up_att = F.interpolate(att, scale_factor=2, mode='bilinear', align_corners=False)
                inserted = box.insert(hr_seg, up_lr.shape[2:], head_os)
                return up_att * inserted + up_lr, defer_logits(hr_seg, (box.h, box.w), fused_ce_consumer(self)), box
            if self.training and not is_teacher:
                box = boxes[0]
                crop_size = (box[1] - box[0], box[3] - box[2])
                mask = lr_seg.new_zeros([nl, 1, *lr_seg.shape[2:]])
                sy, sx = hr_crop_slice(box, 2.0 * head_os)
                mask[:, :, sy, sx] = 1
                att = att * mask
                up_lr = F.interpolate((1 - att) * lr_seg, scale_factor=2, mode='bilinear', align_corners=False)
                up_att = F.interpolate(att, scale_factor=2, mode='bilinear', align_corners=False)
                inserted = torch.zeros_like(up_lr)
                sy, sx = hr_crop_slice(box, head_os)
                inserted[:, :, sy, sx] = hr_seg
                hr_logits = defer_logits(hr_seg, crop_size, fused_ce_consumer(self))
                return up_att * inserted + up_lr, hr_logits, box
            up_lr = F.interpolate((1 - att) * lr_seg, scale_factor=2, mode='bilinear', align_corners=False)
            # overlap-average the sliding crops (the reference rescales `hr_boxes` in place, hrda.py:207-208)
            for i in range(len(boxes)):
                boxes[i] = scale_box(boxes[i], head_os)
            Hh, Ww = max(b[1] for b in boxes), max(b[3] for b in boxes)
            preds = lr_seg.new_zeros((nl, self.num_classes, Hh, Ww))
            count = lr_seg.new_zeros((nl, 1, Hh, Ww))
            for i, (y1, y2, x1, x2) in enumerate(boxes):
                preds[:, :, y1:y2, x1:x2] += hr_seg[i * nl:(i + 1) * nl]
                count[:, :, y1:y2, x1:x2] += 1
            up_att = F.interpolate(att, scale_factor=2, mode='bilinear', align_corners=False)
            return up_att * (preds / count) + up_lr
        return inner
    return deco


# ---------------------------------------------------------------------------------------------------------------------
# loss
# ---------------------------------------------------------------------------------------------------------------------
_FUSED_CE = True
# Whether the consumer of a head's / model's training logits is PixelWeightedCrossEntropyLoss itself (the only thing that
# runs a DeferredUpsample as one kernel) is a property of the OWNING MODEL, carried as an attribute on the module
# (`mark_fused_ce_consumer`): heads used on their own, models with another loss or a subclass that overrides `forward` keep
# getting tensors.  (Round 3 had a module-level flag here: the last model constructed won for every head of the process.)


def mark_fused_ce_consumer(module, loss):
    module._rfn_fused_ce_consumer = type(loss) is PixelWeightedCrossEntropyLoss
    return module._rfn_fused_ce_consumer


def fused_ce_consumer(module):
    return bool(getattr(module, "_rfn_fused_ce_consumer", False))


_CE_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


class DeferredUpsample:
    """Class logits that are still to be up-sampled (bilinear, align_corners=False) to `size` -- what the decode head /
    the training step hand to the loss instead of the up-sampled tensor, so that PixelWeightedCrossEntropyLoss can do
    up-sampling, cross-entropy and their backward in ONE kernel (csrc/loss.hip).  `materialize()` is the tensor the
    reference would have passed (hrda.py:176, segmentation_model.py:163)."""

    def __init__(self, logits, size):
        self.logits, self.size = logits, (int(size[0]), int(size[1]))

    def materialize(self):
        return _up_logits(self.logits, self.size)

    @property
    def shape(self):
        return torch.Size((*self.logits.shape[:2], *self.size))

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        """Any torch function other than the fused loss (F.cross_entropy in a user's own loss, an interpolate, a metric ...)
        gets the tensor the reference would have passed: materialise and go on."""
        conv = lambda a: a.materialize() if isinstance(a, DeferredUpsample) else a  # noqa: E731
        return func(*[conv(a) for a in args], **{k: conv(v) for k, v in (kwargs or {}).items()})


def defer_logits(logits, size, consumer_is_fused_ce=False):
    """-> DeferredUpsample when the consumer is the fused loss and its kernel covers the case (HIP tensor, <= 19 classes,
    scale factors >= 2, gradients wanted), else the up-sampled logits themselves."""
    if _FUSED_CE and consumer_is_fused_ce and logits.is_cuda and logits.dim() == 4 and logits.dtype in _CE_DT and logits.shape[1] <= 19 \
            and size[0] >= 2 * logits.shape[2] and size[1] >= 2 * logits.shape[3] and torch.is_grad_enabled():
        return DeferredUpsample(logits, size)
    return _up_logits(logits, size)


class _UpsampleCEFn(torch.autograd.Function):
    """mean over all pixels of weight * CE(bilinear_up(logits), target); the gradient w.r.t. the low-resolution logits
    comes out of the forward kernel (rfn_upsample_ce) and is scaled by the upstream gradient in backward."""

    @staticmethod
    def forward(ctx, logits, target, weight, size, ignore_index):
        from . import _lib
        from ._tensor import current_stream, on_device, ptr
        B, C, h, w = logits.shape
        H, W = size
        lg = logits.contiguous()
        tg = target.contiguous()
        wt = None if weight is None else weight.to(torch.float32).contiguous()
        if tg.dtype != torch.int64 or tuple(tg.shape) != (B, H, W) or (wt is not None and tuple(wt.shape) != (B, H, W)):
            raise RuntimeError("upsample_ce: target (B, H, W) int64 / weight (B, H, W) expected")
        grad = torch.empty((B, C, h, w), dtype=torch.float32, device=lg.device)
        total = torch.empty(64, dtype=torch.float64, device=lg.device)          # kLossSlots partial sums
        # 16-bit logits: the unfused path stores the up-sampled logits in that dtype before the fp32 softmax
        rc = None
        with on_device(lg.device):
            rc = _lib.load_library().rfn_upsample_ce(ptr(lg), ptr(tg), ptr(wt), ptr(grad), ptr(total), B, C, h, w, H, W,
                                                     int(ignore_index), _CE_DT[lg.dtype], 1, current_stream(lg.device))
        _lib.check(rc, "upsample_ce")
        ctx.save_for_backward(grad)
        ctx.npix, ctx.dtype = float(B * H * W), logits.dtype
        return (total.sum() / ctx.npix).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * (g.to(torch.float32) / ctx.npix)).to(ctx.dtype), None, None, None, None


class PixelWeightedCrossEntropyLoss(nn.Module):
    """models/losses.py:10-22: CE(ignore_index, reduction none) x optional pixel weight, mean over ALL pixels."""

    def __init__(self, ignore_index: int = 255) -> None:
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, input, target, pixel_weight=None):
        if isinstance(input, DeferredUpsample):
            if pixel_weight is not None:
                assert pixel_weight.dim() == target.dim()
            return _UpsampleCEFn.apply(input.logits, target, pixel_weight, input.size, self.ignore_index)
        loss = F.cross_entropy(input, target, ignore_index=self.ignore_index, reduction='none')
        if pixel_weight is not None:
            assert pixel_weight.dim() == loss.dim()
            loss = loss * pixel_weight.to(loss.dtype)
        return loss.mean()
