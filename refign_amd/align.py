"""UAWarpC align path: VGG-16 pyramid -> coarse-to-fine flow + log-variance -> warp of the reference logits.

Host mirror of (brdav/refign):
  models/backbones/vgg.py            VGG(model_type, out_indices, pretrained).forward(x, extract_only_indices)
  models/modules.py:16-56,395-561    ConvBNReLU, OpticalFlowEstimatorResidualConnection, RefinementModule,
                                     UncertaintyModule
  models/heads/uawarpc.py            UAWarpCHead(in_index, input_transform, ..., estimate_uncertainty).forward(
                                     trg, src, trg_256, src_256, out_size)
  models/segmentation_model.py:493-523   align()      models/alignment_model.py:55-79   AlignmentModel.forward

The module tree and therefore the state_dict keys are identical to the reference's
(`decoder4.conv_0.conv.weight`, `estimate_uncertainty_components1.pred_conv_0.bn.running_var`,
`refinement_module_finest.dc_convs.3.conv.weight`, `features.17.bias`, ...) so reference checkpoints load with
strict=True.  The execution is not a transcription:
  * correlation volumes, warps, ReLU/L2-norm and the logits tail run in the HIP kernels of csrc/ -- the warped feature
    maps of levels 3/2/1 are never materialised (warp fused into the correlation staging),
  * the alignment nets are frozen and always in eval() on the UDA step (segmentation_model.py:73-75,693-694), so
    BatchNorm is folded into the convolution weights once and LeakyReLU rides on the conv output,
  * per-channel flow rescalings between pixel-unit systems are folded into scalar multipliers.
Dense convolutions currently go through the ROCm library conv (torch.nn.functional.conv2d -> MIOpen); they are the
next kernels to be replaced by hand-written MFMA implicit-GEMM ("What is library code today").
"""
import math
import os
from typing import List, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import matching
from ._tensor import const_tensor
from .conv import Conv2d
from .layers import LEAKY_SLOPE, ConvBNReLU
from .modules import GlobalFeatureCorrelationLayer, LocalFeatureCorrelationLayer

def _leaky():
    return nn.LeakyReLU(LEAKY_SLOPE)


class OpticalFlowEstimatorResidualConnection(nn.Module):
    """Flow decoder (models/modules.py:395-443): 3x3 stack in->128->128->96->64->32 with two 1x1 skip projections,
    then a 3x3 head to `out_channels`; returns (mapping, 32-channel feature) when output_x."""

    def __init__(self, in_channels, out_channels=2, batch_norm=True, output_x=False, extra_bias='auto'):
        super().__init__()
        self.output_x = output_x
        nl = nn.BatchNorm2d if batch_norm else None
        kw = dict(norm_layer=nl, bias=extra_bias)
        self.conv_0 = ConvBNReLU(in_channels, 128, 3, activation_layer=None, **kw)
        self.conv0_skip = ConvBNReLU(128, 96, 1, norm_layer=nl, activation_layer=None)
        self.conv_1 = ConvBNReLU(128, 128, 3, activation_layer=_leaky, **kw)
        self.conv_2 = ConvBNReLU(128, 96, 3, activation_layer=None, **kw)
        self.conv2_skip = ConvBNReLU(96, 32, 1, norm_layer=nl, activation_layer=None)
        self.conv_3 = ConvBNReLU(96, 64, 3, activation_layer=_leaky, **kw)
        self.conv_4 = ConvBNReLU(64, 32, 3, activation_layer=None, **kw)
        self.predict_mapping = Conv2d(32, out_channels, kernel_size=3, padding=1, bias=True)

    def forward(self, x):
        x0 = self.conv_0(x)                                             # pre-activation kept for the skip
        t = self.conv_2(self.conv_1(F.leaky_relu(x0, LEAKY_SLOPE)))
        x2 = t + self.conv0_skip(x0)
        t = self.conv_4(self.conv_3(F.leaky_relu(x2, LEAKY_SLOPE)))
        feat = F.leaky_relu(t + self.conv2_skip(x2), LEAKY_SLOPE)
        mapping = self.predict_mapping(feat)
        return (mapping, feat) if self.output_x else mapping


class RefinementModule(nn.Module):
    """Dilated context network (models/modules.py:446-477): 32->128->128->128->96->64->32->out, dilations
    1,2,4,8,16,1,1."""

    def __init__(self, in_channels, out_channels=2, batch_norm=True, extra_bias='auto'):
        super().__init__()
        nl = nn.BatchNorm2d if batch_norm else None
        plan = [(in_channels, 128, 1), (128, 128, 2), (128, 128, 4), (128, 96, 8), (96, 64, 16), (64, 32, 1)]
        layers = [ConvBNReLU(ci, co, 3, dilation=d, norm_layer=nl, activation_layer=_leaky, bias=extra_bias)
                  for ci, co, d in plan]
        layers.append(Conv2d(32, out_channels, kernel_size=3, padding=1, bias=True))
        self.dc_convs = nn.Sequential(*layers)

    def forward(self, x):
        return self.dc_convs(x)


class UncertaintyModule(nn.Module):
    """Log-variance head (models/modules.py:480-561).  Front end: every pixel's s x s correlation patch is a
    1-channel micro-image run through valid 3x3 convs (s=9: 9->7->5->3->1; s=16: 16->14->maxpool 7->5->3->1) giving 6
    numbers per pixel; back end: cat(6, 32-ch decoder feature[, previous log-var, previous flow]) -> 32 -> 16 -> 1."""

    def __init__(self, in_channels, feed_in_previous=False, out_channels=1, search_size=9, batch_norm=True,
                 depthwise_separable=False):
        super().__init__()
        if depthwise_separable:
            raise NotImplementedError
        if search_size not in (9, 16):
            raise ValueError("search_size must be 9 or 16")
        nl = nn.BatchNorm2d if batch_norm else None
        self.search_size = search_size
        self.feed_in_previous = feed_in_previous
        add = 3 if feed_in_previous else 0
        kw = dict(kernel_size=3, stride=1, padding=0, norm_layer=nl, activation_layer=_leaky)
        self.conv_0 = ConvBNReLU(in_channels, 32, **kw)
        if search_size == 16:
            self.maxpool = nn.MaxPool2d((2, 2))
        self.conv_1 = ConvBNReLU(32, 32, **kw)
        self.conv_2 = ConvBNReLU(32, 16, **kw)
        self.predict_uncertainty = Conv2d(16, 6, kernel_size=3, stride=1, padding=0, bias=True)
        self.pred_conv_0 = ConvBNReLU(6 + 32 + add, 32, 3, norm_layer=nl, activation_layer=_leaky)
        self.pred_conv_1 = ConvBNReLU(32, 16, 3, norm_layer=nl, activation_layer=_leaky)
        self.predict_uncertainty_final = Conv2d(16, 1, kernel_size=3, stride=1, padding=1, bias=True)
        self._packed = None

    # micro-images per launch of the library conv chain: bounds the (N,32,7,7) intermediates to ~0.8 GB
    MICRO_BATCH = 131072

    def packed_frontend_weights(self):
        """The four front-end convs with eval-mode BatchNorm folded in, packed for rfn_uncertainty9_frontend_f32
        (layout: include/refign_hip.h): W[k][n] with k = (ky*3+kx)*Cin + ci.  Cached until train()/load/_apply."""
        if self._packed is None:
            parts = []
            for m in (self.conv_0, self.conv_1, self.conv_2):
                w, b = m.folded()
                parts += [w.permute(2, 3, 1, 0).reshape(-1), b.reshape(-1)]
            pu = self.predict_uncertainty
            parts += [pu.weight.detach().permute(2, 3, 1, 0).reshape(-1), pu.bias.detach().reshape(-1)]
            self._packed = torch.cat([t.float() for t in parts]).contiguous()
        return self._packed

    def train(self, mode=True):
        self._packed = None
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._packed = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def patch_statistics(self, corr):
        """corr (B, s*s, H, W) -> (B, 6, H, W).  Search size 9 in eval mode on the GPU (the frozen alignment head:
        every call of the hot path) is ONE fused HIP kernel; training mode / search size 16 (the 16x16 level-4 map,
        256 micro-images per image) run the library conv chain."""
        b, _, h, w = corr.shape
        s = self.search_size
        if (s == 9 and corr.is_cuda and not self.training and not torch.is_grad_enabled()
                and self.conv_0.use_norm and corr.dtype == torch.float32
                and os.environ.get("RFN_UNCERT_FUSED", "1") != "0"):
            return matching.uncertainty9_frontend(corr, self.packed_frontend_weights(),
                                                  half_matrix=_HEAD_IN_TIMED_MAP[0] or align_compute_dtype() != torch.float32)
        if s == 9 and (self.training or torch.is_grad_enabled()):
            return self._patch_statistics_tiled(corr)
        x = corr.permute(0, 2, 3, 1).reshape(b * h * w, 1, s, s)
        outs = []
        for i in range(0, x.shape[0], self.MICRO_BATCH):
            y = self.conv_0(x[i:i + self.MICRO_BATCH])
            if s == 16:
                y = self.maxpool(y)
            y = self.predict_uncertainty(self.conv_2(self.conv_1(y)))
            outs.append(y.flatten(1))
        y = outs[0] if len(outs) == 1 else torch.cat(outs)
        return y.view(b, h, w, 6).permute(0, 3, 1, 2)

    def _patch_statistics_tiled(self, corr):
        """The micro-image chain under autograd (matcher training) as FOUR convolutions on
        ordinary images instead of one convolution per 100 000 9x9 micro-images (which the library runs as one
        im2col + GEMM pair per micro-image: 35 000 launches per call, measured): the micro-image of pixel (y, x) is tile
        (y, x) of a (9 h, 9 w) image; a valid 3x3 convolution of the tiled image contains every micro-image's 7x7
        result (plus windows that straddle tiles, dropped); the 7x7 results are re-tiled into a (7 h, 7 w) image, and so
        on 9 -> 7 -> 5 -> 3 -> 1.  BatchNorm over the re-tiled image sees exactly the N x 49 (25, 9) values per
        channel it sees in the micro-image batch, so batch statistics and running buffers are the reference's."""
        b, _, h, w = corr.shape

        def retile(y, k):                                # (b, C, k h + 2 - 2.., ..): valid part of every (k+2)-tile
            r = matching.retile_valid(y, h, w, k)        # one gather kernel on channels-last memory (csrc/warp.hip)
            if r is not None:
                return r
            y = F.pad(y, (0, 2, 0, 2))
            c = y.shape[1]
            return y.view(b, c, h, k + 2, w, k + 2)[:, :, :, :k, :, :k].reshape(b, c, k * h, k * w)

        from . import bn as bnk
        from .params import compute_dtype
        x = corr.view(b, 9, 9, h, w).permute(0, 3, 1, 4, 2).reshape(b, 1, 9 * h, 9 * w)
        for m, k in ((self.conv_0, 7), (self.conv_1, 5), (self.conv_2, 3)):
            x = retile(m._conv2d(x, m.conv.weight, m.conv.bias), k)
            cd = compute_dtype(x)
            if m.use_norm and m.act == 'leaky' and m.act_slope == LEAKY_SLOPE and x.is_cuda and \
                    os.environ.get("RFN_BN_KERNEL", "1") != "0" and bnk.usable(x, m.bn, cd):
                x = bnk.bn_act_train(x, m.bn, 3, cd)         # BatchNorm(train) + LeakyReLU in two passes (csrc/bn.hip)
                continue
            if m.use_norm:
                x = m.bn(x)
            x = F.leaky_relu(x, m.act_slope, inplace=True) if m.act == 'leaky' else F.relu(x, inplace=True)
        y = self.predict_uncertainty(x)                                                    # (b, 6, 3 h - 2, 3 w - 2)
        return F.pad(y, (0, 2, 0, 2)).view(b, 6, h, 3, w, 3)[:, :, :, 0, :, 0]

    def forward(self, corr, feat, up_previous_uncertainty=None, up_previous_flow=None):
        parts = [self.patch_statistics(corr), feat]
        if self.feed_in_previous:
            parts += [up_previous_uncertainty, up_previous_flow]
        u = self.pred_conv_1(self.pred_conv_0(parts))
        return self.predict_uncertainty_final(u)


# ---------------------------------------------------------------------------------------------------------------------
# VGG-16 pyramid
# ---------------------------------------------------------------------------------------------------------------------
_VGG_CFG = {
    "vgg11": [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
    "vgg13": [64, 64, "M", 128, 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
    "vgg16": [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"],
    "vgg19": [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"],
}


_POOL_KERNEL = True


def _maxpool2x2(x, m):
    """nn.MaxPool2d(2, 2) (floor mode) of an NCHW-shaped tensor with channels-last 16-bit memory on the HIP kernel
    (csrc/warp.hip rfn_maxpool2x2_nhwc16); None outside that domain."""
    def two(v):
        return v == 2 or tuple(v) == (2, 2) if not isinstance(v, int) else v == 2
    if not (two(m.kernel_size) and two(m.stride) and m.padding in (0, (0, 0)) and m.dilation in (1, (1, 1))
            and not m.ceil_mode and not m.return_indices):
        return None
    if x.dim() != 4 or x.dtype not in (torch.bfloat16, torch.float16) or x.shape[1] % 8 or x.shape[2] < 2 or x.shape[3] < 2:
        return None
    xh = x.permute(0, 2, 3, 1)
    if not xh.is_contiguous():
        return None
    from . import _lib
    from ._tensor import current_stream, on_device, ptr
    B, H, W, C = xh.shape
    y = torch.empty((B, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
    with on_device(x.device):
        rc = _lib.load_library().rfn_maxpool2x2_nhwc16(ptr(xh), ptr(y), B, H, W, C, 1 if x.dtype == torch.bfloat16 else 2,
                                                       current_stream(x.device))
    _lib.check(rc, "maxpool2x2_nhwc16")
    return y.permute(0, 3, 1, 2)


class VGG(nn.Module):
    """models/backbones/vgg.py:33-149.  `features` is an nn.Sequential with the torchvision layout (so
    `features.N.weight` keys match); tap points are after the first ReLU and after every max-pool; `out_indices`
    selects taps and `extract_only_indices` sub-selects them with early exit."""

    def __init__(self, model_type: str, out_indices: list = [0, 1, 2, 3, 4, 5], pretrained: Optional[str] = None):
        super().__init__()
        self.model_type = model_type
        bn = model_type.endswith("_bn")
        layers, taps, cin, first = [], [], 3, True
        for v in _VGG_CFG[model_type.replace("_bn", "")]:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
                taps.append(len(layers))
            else:
                layers.append(Conv2d(cin, v, kernel_size=3, padding=1))
                if bn:
                    layers.append(nn.BatchNorm2d(v))
                layers.append(nn.ReLU(inplace=True))
                cin = v
                if first:
                    taps.append(len(layers))
                    first = False
        self.features = nn.Sequential(*layers)
        self.layer_indices = [taps[i] for i in out_indices]
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.zeros_(m.bias)
        if pretrained is not None:
            self.load_weights(pretrained)

    def load_weights(self, pretrained):
        """vgg.py:91-106: path or 'imagenet' (needs the torchvision checkpoint on disk: no network here); drops
        classifier.* keys; strict."""
        if pretrained == 'imagenet':
            pretrained = os.path.join(os.environ.get('TORCH_HOME', os.path.expanduser('~/.cache/torch')), 'hub',
                                      'checkpoints', 'vgg16-397923af.pth')
        if not os.path.exists(pretrained):
            raise FileNotFoundError(f"VGG weights not found: {pretrained}")
        ckpt = torch.load(pretrained, map_location='cpu')
        sd = ckpt.get('state_dict', ckpt)
        self.load_state_dict({k: v for k, v in sd.items() if not k.startswith('classifier.')}, strict=True)

    def _run(self, x, lo, hi):
        """features[lo:hi] (conv3x3 + bias + ReLU as one launch of the implicit-GEMM kernel)."""
        i = lo
        while i < hi:
            m = self.features[i]
            if isinstance(m, nn.Conv2d) and i + 1 < hi and isinstance(self.features[i + 1], nn.ReLU) and x.is_cuda \
                    and not torch.is_grad_enabled():
                # conv3x3 + bias + ReLU: one launch of the hand-written implicit-GEMM kernel (refign_amd/conv.py)
                from .conv import conv2d_mfma
                from .params import compute_dtype
                y = conv2d_mfma(x, m.weight, m.bias, m.stride, m.padding, m.dilation, act='relu', dtype=compute_dtype(x))
                if y is not None:
                    x = y
                    i += 2
                    continue
            if isinstance(m, nn.MaxPool2d) and x.is_cuda and not torch.is_grad_enabled() and _POOL_KERNEL:
                y = _maxpool2x2(x, m)
                if y is not None:
                    x = y
                    i += 1
                    continue
            x = m(x)
            i += 1
        return x

    def forward(self, x, extract_only_indices=None):
        idx = [self.layer_indices[i] for i in extract_only_indices] if extract_only_indices else self.layer_indices
        outs, prev = [], 0
        for i in idx:
            x = self._run(x, prev, i)
            outs.append(x)
            prev = i
        return outs


# ---------------------------------------------------------------------------------------------------------------------
# UAWarpC head
# ---------------------------------------------------------------------------------------------------------------------
def _up(x, size):
    return F.interpolate(x, size=size, mode='bilinear', align_corners=False)


class BaseHead(nn.Module):
    """models/heads/base.py:7-44."""

    def __init__(self, num_classes, in_index, input_transform=None):
        super().__init__()
        self.input_transform = input_transform
        self.in_index = in_index[0] if isinstance(in_index, (list, tuple)) and len(in_index) == 1 else in_index
        self.num_classes = num_classes

    def _transform_inputs(self, inputs):
        if self.input_transform == 'resize_concat':
            inputs = [inputs[i] for i in self.in_index]
            return torch.cat([_up(x, inputs[0].shape[2:]) for x in inputs], dim=1)
        if self.input_transform == 'multiple_select':
            return [inputs[i] for i in self.in_index]
        return inputs[self.in_index]


class UAWarpCHead(BaseHead):
    """models/heads/uawarpc.py:17-280.  forward(trg, src, trg_256, src_256, out_size) returns the four
    (flow, log-variance) pairs, coarse to fine; flows are in pixels of the ORIGINAL resolution `out_size`."""

    def __init__(self, in_index: Union[List[int], int], input_transform: Optional[str] = None,
                 pretrained: Optional[str] = None, batch_norm: bool = True, refinement_at_adaptive_res: bool = True,
                 refinement_at_finest_level: bool = True, estimate_uncertainty: bool = False,
                 uncertainty_mixture: bool = False, iterative_refinement: bool = False):
        super().__init__(None, in_index, input_transform)
        self.estimate_uncertainty = estimate_uncertainty
        self.uncertainty_mixture = uncertainty_mixture
        self.iterative_refinement = iterative_refinement
        self.refinement_at_adaptive_res = refinement_at_adaptive_res
        self.refinement_at_finest_level = refinement_at_finest_level
        self.global_corr = GlobalFeatureCorrelationLayer(cyclic_consistency=True)
        self.local_corr = LocalFeatureCorrelationLayer(patch_size=9)
        u = 1 if estimate_uncertainty else 0
        mk = lambda cin: OpticalFlowEstimatorResidualConnection(cin, batch_norm=batch_norm, output_x=True)  # noqa: E731
        self.decoder4 = mk(16 * 16)
        self.decoder3 = mk(81 + 2 + u)
        if refinement_at_adaptive_res:
            self.refinement_module_adaptive = RefinementModule(32, batch_norm=batch_norm)
        self.decoder2 = mk(81 + 2 + u)
        self.reduce = Conv2d(32, 2, kernel_size=1, bias=True)
        self.decoder1 = mk(81 + 2 + 2 + u)
        if refinement_at_finest_level:
            self.refinement_module_finest = RefinementModule(32, batch_norm=batch_norm)
        if estimate_uncertainty:
            self.estimate_uncertainty_components4 = UncertaintyModule(1, search_size=16)
            for lvl in (3, 2, 1):
                setattr(self, f"estimate_uncertainty_components{lvl}",
                        UncertaintyModule(1, search_size=9, feed_in_previous=True))
        if pretrained is not None:
            self.load_weights(pretrained)

    # -- one coarse-to-fine refinement level -------------------------------------------------------------------------
    def _level(self, lvl, feat_trg, feat_src, flow_prev, uncert_prev, orig_size, extra=None):
        """flow_prev / uncert_prev are already up-sampled to this level's size; flow_prev is in the pixel units the
        decoders work in (256-space at level 3, original pixels at levels 2 and 1) and `orig_size` is the size those
        units refer to.  Returns (residual-corrected flow, decoder feature, log-variance)."""
        h, w = feat_trg.shape[-2:]
        oh, ow = orig_size
        # flows / log-variances travel between levels in fp32 whatever dtype the convolutions ran in: half precision
        # resolves 1 px at |flow| >= 1024 px
        flow_prev = flow_prev.float()
        uncert_prev = None if uncert_prev is None else uncert_prev.float()
        scale = const_tensor([w / float(ow), h / float(oh)], flow_prev).view(1, 2, 1, 1)
        corr = self.local_corr(feat_src, feat_trg, flow=(flow_prev * scale).contiguous())   # warp fused in
        parts = [corr, flow_prev] + ([extra] if extra is not None else [])
        if self.estimate_uncertainty:
            parts.append(uncert_prev)
        res, x = getattr(self, f"decoder{lvl}")(parts)        # (the concatenation as its parts: layers.ConvBNReLU.forward)
        refine = {3: self.refinement_at_adaptive_res and 'refinement_module_adaptive',
                  1: self.refinement_at_finest_level and 'refinement_module_finest'}.get(lvl)
        if refine:
            res = res.float() + getattr(self, refine)(x).float()
        flow = res.float() + flow_prev
        uncert = None
        if self.estimate_uncertainty:
            uncert = getattr(self, f"estimate_uncertainty_components{lvl}")(corr, x, uncert_prev, flow_prev).float()
        return flow, x, uncert

    def forward(self, trg, src, trg_256, src_256, out_size):
        c11, c12 = self._transform_inputs(trg)
        c13, c14 = self._transform_inputs(trg_256)
        c21, c22 = self._transform_inputs(src)
        c23, c24 = self._transform_inputs(src_256)
        c11, c12, c13, c14, c21, c22, c23, c24 = [matching.l2_normalize_channels(c)
                                                  for c in (c11, c12, c13, c14, c21, c22, c23, c24)]
        H, W = out_size
        eu = self.estimate_uncertainty

        # level 4: 16x16 global correlation, mapping regression (uawarpc.py:111-130)
        assert tuple(c14.shape[-2:]) == (16, 16), tuple(c14.shape[-2:])
        corr4 = self.global_corr(c24, c14)
        est_map4, x4 = self.decoder4(corr4)
        flow4_256 = matching.unnormalise_and_convert_mapping_to_flow(est_map4.float()) * (256.0 / 16.0)
        u4_256 = None
        if eu:
            u4_256 = self.estimate_uncertainty_components4(corr4, x4).float()
            u4_256 = u4_256 + 2 * math.log(256.0 / 16.0)

        # level 3: 32x32, decoders work in 256-space pixels (uawarpc.py:132-173)
        assert tuple(c13.shape[-2:]) == (32, 32), tuple(c13.shape[-2:])
        up_flow4 = _up(flow4_256, (32, 32))
        up_u4 = _up(u4_256, (32, 32)) if eu else None
        flow3, x3, u3 = self._level(3, c13, c23, up_flow4, up_u4, (256, 256))
        flow3 = flow3 * const_tensor([W / 256.0, H / 256.0], flow3).view(1, 2, 1, 1)
        diag_term = 2 * math.log(math.sqrt(H ** 2 + W ** 2) / math.sqrt(2 * 256.0 ** 2))
        if eu:
            u3 = u3 + diag_term
        if self.iterative_refinement and not self.training:
            # uawarpc.py:175-207 (set in the megadepth configs, unset in every refign_* config): for images of 1086 pixels
            # or more the jump from the 32x32 level to 1/8 resolution is bridged by extra passes of the LEVEL-2 decoder
            # at 1/16, 1/32, ... resolution on area-down-sampled level-2 features, so that no up-sampling step exceeds 2x.
            R = float(max(H, W)) / 8.0 / 32.0
            extra = max(0, int(round(math.log(R / 3.0) / math.log(2))))
            for n in range(extra):
                ratio = 1.0 / (8.0 * 2 ** (extra - n))
                size = (int(H * ratio), int(W * ratio))
                up_flow, up_u = _up(flow3, size), (_up(u3, size) if eu else None)
                c2s = F.interpolate(c22, size=size, mode='area')
                c1s = F.interpolate(c12, size=size, mode='area')
                corr = self.local_corr(c2s, c1s, flow=(up_flow * ratio).contiguous())
                res, x3b = self.decoder2(torch.cat([corr, up_flow] + ([up_u] if eu else []), 1))
                flow3 = res.float() + up_flow
                if eu:
                    u3 = self.estimate_uncertainty_components2(corr, x3b, up_u, up_flow).float()

        # level 2: 1/8 resolution, original pixel units (uawarpc.py:209-234)
        s2 = c12.shape[-2:]
        flow2, x2, u2 = self._level(2, c12, c22, _up(flow3, s2), _up(u3, s2) if eu else None, (H, W))

        # level 1: 1/4 resolution (uawarpc.py:236-271)
        s1 = c11.shape[-2:]
        up_feat2 = self.reduce(_up(x2, s1))
        flow1, _, u1 = self._level(1, c11, c21, _up(flow2, s1), _up(u2, s1) if eu else None, (H, W), extra=up_feat2)

        flow4 = flow4_256 * const_tensor([W / 256.0, H / 256.0], flow4_256).view(1, 2, 1, 1)
        if eu:
            return (flow4, u4_256 + diag_term), (flow3, u3), (flow2, u2), (flow1, u1)
        return flow4, flow3, flow2, flow1

    def load_weights(self, pretrain_path):
        """uawarpc.py:282-305: accepts a Lightning checkpoint and keeps the `alignment_head.` sub-tree; strict."""
        if pretrain_path is None:
            return
        for cand in (pretrain_path, os.path.join(os.environ.get('TORCH_HOME', ''), 'hub', pretrain_path)):
            if os.path.exists(cand):
                ckpt = torch.load(cand, map_location='cpu')
                break
        else:
            raise FileNotFoundError(f"UAWarpC weights not found: {pretrain_path} (no network access)")
        sd = ckpt.get('state_dict', ckpt)
        self.load_state_dict({k[len('alignment_head.'):]: v for k, v in sd.items()
                              if k.startswith('alignment_head.')}, strict=True)


# ---------------------------------------------------------------------------------------------------------------------
# align(): images + reference logits -> warped logits, validity mask, confidence
# ---------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def extract_pyramids(alignment_backbone, images_ref, images_trg):
    """segmentation_model.py:497-510: full-resolution pyramid (1/4, 1/8) and the 256x256 pyramid (32^2, 16^2) for
    the concatenated (ref, trg) batch."""
    b = images_trg.shape[0]
    ref_256 = matching.area_resize(images_ref, (256, 256))
    trg_256 = matching.area_resize(images_trg, (256, 256))
    feats = alignment_backbone(torch.cat([images_ref, images_trg]), extract_only_indices=[-3, -2])
    feats_256 = alignment_backbone(torch.cat([ref_256, trg_256]), extract_only_indices=[-2, -1])
    pyr_ref, pyr_trg = zip(*[torch.split(f, [b, b]) for f in feats])
    pyr_ref_256, pyr_trg_256 = zip(*[torch.split(f, [b, b]) for f in feats_256])
    return pyr_trg, pyr_ref, pyr_trg_256, pyr_ref_256


@torch.no_grad()
def align(alignment_backbone, alignment_head, logits_ref, images_ref, images_trg):
    """DomainAdaptationSegmentationModel.align (segmentation_model.py:493-523).
    Returns (warped_ref_logits (b,C,h,w), trg_ref_mask (b,h,w) bool, trg_ref_cert (b,1,h,w))."""
    h, w = images_trg.shape[-2:]
    if tuple(logits_ref.shape[-2:]) != (h, w):
        raise RuntimeError("align: logits_ref must have the image resolution")
    # Precision map of the reference under its AMP recipe (`--trainer.precision 16`, README.md:262): the VGG and decoder
    # convolutions run in fp16 autocast, correlation and warp are forced to fp32 (correlation_function.py:51,
    # matching_utils.py:40-43).  Same here: inside a reduced-precision autocast region the matcher's convolutions run
    # in fp16 (the reference's own dtype -- not bf16, sub-pixel flow accuracy rests on the mantissa), the HIP
    # correlation / warp / L2-norm / uncertainty kernels always compute in fp32; outside autocast (fp32 parity mode)
    # everything is fp32.  RFN_ALIGN_DTYPE=fp32|fp16|bf16 overrides.
    dt = align_compute_dtype()
    with torch.autocast("cuda", enabled=dt != torch.float32, dtype=dt if dt != torch.float32 else None):
        pyr = extract_pyramids(alignment_backbone, images_ref.float(), images_trg.float())
        flow_q, logvar_q = run_head(alignment_head, pyr, (h, w))[-1]
        return matching.align_tail(logits_ref, flow_q.float(), logvar_q.float())


# The timed precision map of the matcher (round 6).  Inside the step's 16-bit autocast region the VGG-16 pyramid runs in fp16
# (the reference's own AMP dtype) and is accurate enough: with the head in fp32 the warped logits at 1080 x 1920 sit within 8e-5 of
# the reference's fp32 CPU path.  The HEAD is where fp16 loses the north star's 1e-3: its decoders take the flow itself as input
# channels -- hundreds of pixels at full resolution, where fp16 resolves 0.125-0.25 px -- and carry activations of that
# magnitude through 14 layers; input, weight and output rounding contribute alike (profiles/r06_align_precision_layers.txt:
# flow error 0.07 / 0.11 / 0.11 / 0.19 px after levels 4 / 3 / 2 / 1, warped logits 5.6e-3).  So the head's convolutions run as
# split-bf16 products (three MFMA products per convolution, fp32 activations between the layers; refign_amd/split32.py) --
# 2^-16 relative instead of 2^-11 -- while the per-pixel 9 x 9 micro-image front end of the uncertainty module keeps its f16
# matrix layers (inputs in [0, 1], fp32 elsewhere; 5e-4).  HEAD_SPLIT = False: the head in the autocast dtype (round 5).
HEAD_SPLIT = True
_HEAD_IN_TIMED_MAP = [False]


def run_head(alignment_head, pyr, size):
    """alignment_head(*pyr, size) under the precision map above."""
    dt = align_compute_dtype()
    if HEAD_SPLIT and dt != torch.float32 and pyr[0][0].is_cuda and not torch.is_grad_enabled():
        was = _HEAD_IN_TIMED_MAP[0]
        _HEAD_IN_TIMED_MAP[0] = True
        try:
            with torch.autocast("cuda", enabled=False):
                # (the fp16 pyramid as it is: the head's first step is the L2 normalisation, whose kernel takes channels-last 16-bit
                # maps and writes fp32 NCHW in one pass -- everything after it is fp32)
                return alignment_head(*pyr, size)
        finally:
            _HEAD_IN_TIMED_MAP[0] = was
    return alignment_head(*pyr, size)


def align_flow(alignment_backbone, alignment_head, images_ref, images_trg):
    """The image-only part of align(): quarter-resolution flow and log-variance of the mixture density (fp32).  It depends on
    the two images and on the frozen matcher, on nothing that training changes -- uda.prefetch_align_flow computes it for
    the NEXT batch while this step's mixed pass has the device mostly to itself."""
    h, w = images_trg.shape[-2:]
    dt = align_compute_dtype()
    with torch.autocast("cuda", enabled=dt != torch.float32, dtype=dt if dt != torch.float32 else None):
        pyr = extract_pyramids(alignment_backbone, images_ref.float(), images_trg.float())
        flow_q, logvar_q = run_head(alignment_head, pyr, (h, w))[-1]
    return flow_q.float(), logvar_q.float()


def align_from_flow(logits_ref, flow_q, logvar_q):
    """The rest of align(): (warped_ref_logits, mask, certainty) from the reference logits and align_flow()'s result."""
    return matching.align_tail(logits_ref, flow_q, logvar_q)


_ALIGN_DTYPES = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def align_compute_dtype():
    env = os.environ.get("RFN_ALIGN_DTYPE")
    if env:
        return _ALIGN_DTYPES[env]
    return torch.float16 if torch.is_autocast_enabled("cuda") else torch.float32


@torch.no_grad()
def alignment_forward(alignment_backbone, alignment_head, images_i, images_j):
    """AlignmentModel.forward (alignment_model.py:55-79): flow i->j at full resolution and 1 - P_R."""
    h, w = images_i.shape[-2:]
    pyr = extract_pyramids(alignment_backbone, images_j, images_i)
    flow_q, logvar_q = run_head(alignment_head, pyr, (h, w))[-1]
    flow = _up(flow_q, (h, w))
    uncert = _up(logvar_q, (h, w))
    return flow, 1.0 - matching.estimate_probability_of_confidence_interval_of_mixture_density(uncert, R=1.0)
