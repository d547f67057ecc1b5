"""Host mirror of helpers/matching_utils.py for the functions on the hot path (warp, mapping->flow, confidence),
backed by the HIP kernels in csrc/warp.hip.  Same names, argument meaning and return conventions."""
import os

import torch

from . import _lib
from ._tensor import current_stream, ptr, require_device_tensor, same_device, on_device


def warp(x, flo, padding_mode='zeros', return_mask=False):
    """warp an image/tensor back according to the flow (matching_utils.py:11-49).

    x: [B,C,H,W], flo: [B,2,H,W] in pixels.  Bilinear, align_corners=True, zero padding; the mask is True where the
    normalised sampling position is strictly inside (-1,1)^2.  Like the reference, an identically-zero flow returns
    `x` itself (and an all-True mask) -- that costs one host sync, as it does there (matching_utils.py:19).
    """
    if padding_mode != 'zeros':
        raise RuntimeError("warp: only padding_mode='zeros' is used by the reference hot path")
    x = require_device_tensor(x.float().contiguous(), "x", torch.float32)
    flo = require_device_tensor(flo.float().contiguous(), "flo", torch.float32)
    dev = same_device(x, flo)
    B, C, H, W = x.shape
    if tuple(flo.shape) != (B, 2, H, W):
        raise RuntimeError("warp: flo must be (B,2,H,W)")
    if bool(torch.all(flo == 0)):
        if return_mask:
            return x, torch.ones((B, H, W), dtype=torch.bool, device=dev)
        return x
    if torch.is_grad_enabled() and (x.requires_grad or flo.requires_grad):
        # matcher training (alignment_model.py:81-146): differentiable in the features AND in the flow, like grid_sample
        out = _WarpFn.apply(x, flo)
        if return_mask:
            return out, warp_nocheck(x.detach(), flo.detach(), True)[1]
        return out
    return warp_nocheck(x, flo, return_mask)


class _WarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flo):
        ctx.save_for_backward(x, flo)
        return warp_nocheck(x, flo)

    @staticmethod
    def backward(ctx, g):
        x, flo = ctx.saved_tensors
        g = g.float().contiguous()
        B, C, H, W = x.shape
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gf = torch.empty_like(flo) if ctx.needs_input_grad[1] else None
        lib = _lib.load_library()
        with on_device(x.device):
            rc = lib.rfn_warp_bwd_f32(ptr(x), ptr(flo), ptr(g), ptr(gx), ptr(gf), B, C, H, W, current_stream(x.device))
        _lib.check(rc, "warp backward")
        return gx, gf


def warp_nocheck(x, flo, return_mask=False):
    """warp() without the `flo == 0` host synchronisation (graph-capturable)."""
    dev = x.device
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    mask = torch.empty((B, H, W), dtype=torch.uint8, device=dev) if return_mask else None
    lib = _lib.load_library()
    with on_device(dev):
        rc = lib.rfn_warp_f32(ptr(x), ptr(flo), ptr(out), ptr(mask), B, C, H, W, current_stream(dev))
    _lib.check(rc, "warp")
    if return_mask:
        return out, mask.view(torch.bool)
    return out


_DT16 = {torch.bfloat16: 1, torch.float16: 2}


def l2_normalize_channels(x):
    """F.normalize(x.float(), p=2, dim=1) for (B, C, H, W) features (uawarpc.py:101-108) -> NCHW float32.  Channels-last
    16-bit features (what the matcher's convolutions deliver under the AMP recipe) are widened, re-laid out and
    normalised in one kernel; anything else is made NCHW float32 first."""
    lib = _lib.load_library()
    if x.is_cuda and x.dim() == 4 and x.dtype in _DT16 and x.shape[1] % 8 == 0 and x.shape[1] <= 2048 and \
            x.is_contiguous(memory_format=torch.channels_last):
        B, C, H, W = x.shape
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
        with on_device(x.device):
            rc = lib.rfn_l2norm_channels_nhwc16_f32(ptr(x), ptr(out), B, C, H * W, _DT16[x.dtype], current_stream(x.device))
        _lib.check(rc, "l2_normalize_channels (channels-last 16-bit)")
        return out
    x = require_device_tensor(x.float().contiguous(), "x", torch.float32)
    B, C = x.shape[:2]
    hw = x[0, 0].numel()
    out = torch.empty_like(x)
    with on_device(x.device):
        rc = lib.rfn_l2norm_channels_f32(ptr(x), ptr(out), B, C, hw, current_stream(x.device))
    _lib.check(rc, "l2_normalize_channels")
    return out


def uncertainty9_frontend(corr, packed_weights, half_matrix=False):
    """Front end of UncertaintyModule (search size 9, eval mode; models/modules.py:529-551): (B,81,H,W) correlation
    volume -> (B,6,H,W), the 9x9 -> 7x7 -> 5x5 -> 3x3 -> 1x1 micro-conv chain of every pixel in one HIP kernel.
    `packed_weights`: UncertaintyModule.packed_frontend_weights() (BatchNorm folded, layout in refign_hip.h)."""
    corr = require_device_tensor(corr.contiguous(), "corr", torch.float32)
    w = require_device_tensor(packed_weights.contiguous(), "packed_weights", torch.float32)
    B, D, H, W = corr.shape
    lib = _lib.load_library()
    if D != 81 or w.numel() != lib.rfn_uncertainty9_weights_len():
        raise RuntimeError("uncertainty9_frontend: corr must be (B,81,H,W) and the weight pack %d floats"
                           % lib.rfn_uncertainty9_weights_len())
    same_device(corr, w)
    out = torch.empty((B, 6, H, W), dtype=torch.float32, device=corr.device)
    with on_device(corr.device):
        # half_matrix: the 32 -> 32 and 32 -> 16 layers on the f16 matrix pipe (the reference's AMP precision for these convolutions)
        fn = lib.rfn_uncertainty9_frontend_f16mm if half_matrix else lib.rfn_uncertainty9_frontend_f32
        rc = fn(ptr(corr), ptr(w), ptr(out), B, H, W, current_stream(corr.device))
    _lib.check(rc, "uncertainty9_frontend")
    return out


class _RetileFn(torch.autograd.Function):
    """csrc/warp.hip rfn_retile_copy under autograd: y (B, C, (k+2) h - 2, (k+2) w - 2) with channels-last memory (pixels `S` elements
    apart, S >= C: the convolution kernels pad their output channels) -> (B, C, k h, k w), channels-last."""

    @staticmethod
    def forward(ctx, y, h, w, k):
        B, C, Hs, Ws = y.shape
        ctx.geom = (B, h, w, k, C)
        es = y.element_size()
        out = torch.empty((B, k * h, k * w, C), dtype=y.dtype, device=y.device)
        with on_device(y.device):
            rc = _lib.load_library().rfn_retile_copy(ptr(y), ptr(out), B, h, w, k, C * es // 16, y.stride(3) * es // 16, 0,
                                                     current_stream(y.device))
        _lib.check(rc, "retile_copy")
        return out.permute(0, 3, 1, 2)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go):
        B, h, w, k, C = ctx.geom
        gh = go.permute(0, 2, 3, 1).contiguous()
        u = C * gh.element_size() // 16
        gy = torch.empty((B, (k + 2) * h - 2, (k + 2) * w - 2, C), dtype=gh.dtype, device=gh.device)
        with on_device(gh.device):
            rc = _lib.load_library().rfn_retile_copy(ptr(gh), ptr(gy), B, h, w, k, u, u, 1, current_stream(gh.device))
        _lib.check(rc, "retile_copy (backward)")
        return gy.permute(0, 3, 1, 2), None, None, None


def retile_valid(y, h, w, k):
    """The k x k valid part of every (k + 2) x (k + 2) tile of an NCHW-shaped map y (B, C, (k+2) h - 2, (k+2) w - 2), re-tiled into
    (B, C, k h, k w) -- one gather kernel on channels-last memory, differentiable (align.py UncertaintyModule._patch_statistics_tiled).
    None outside the kernel's domain (not on a GPU, not channels-last rows of whole 16-byte units): the caller pads, views and slices
    instead."""
    if not y.is_cuda or y.dim() != 4 or y.shape[2] != (k + 2) * h - 2 or y.shape[3] != (k + 2) * w - 2:
        return None
    B, C, Hs, Ws = y.shape
    es, S = y.element_size(), y.stride(3)
    if y.stride(1) != 1 or S < C or y.stride(2) != Ws * S or y.stride(0) != Hs * Ws * S or (C * es) % 16 or (S * es) % 16 \
            or y.data_ptr() % 16:
        return None
    return _RetileFn.apply(y, h, w, k)


def area_resize(x, size):
    """F.interpolate(x, size=size, mode='area') (segmentation_model.py:498-501) on the HIP kernel."""
    x = require_device_tensor(x.float().contiguous(), "x", torch.float32)
    B, C, H, W = x.shape
    out = torch.empty((B, C, size[0], size[1]), dtype=torch.float32, device=x.device)
    lib = _lib.load_library()
    with on_device(x.device):
        rc = lib.rfn_area_resize_f32(ptr(x), ptr(out), B * C, H, W, size[0], size[1], current_stream(x.device))
    _lib.check(rc, "area_resize")
    return out


def unnormalise_and_convert_mapping_to_flow(map, output_channel_first=True):
    """matching_utils.py:77-103 (4-D case): mapping normalised to [-1,1] -> flow in pixels.  16x16 only on the hot
    path, so this stays a handful of tensor ops."""
    if map.dim() != 4:
        raise RuntimeError("unnormalise_and_convert_mapping_to_flow: expects (B,2,H,W)")
    if map.shape[1] != 2:
        map = map.permute(0, 3, 1, 2)
    B, C, H, W = map.shape
    xx = torch.arange(0, W, dtype=map.dtype, device=map.device).view(1, 1, W)
    yy = torch.arange(0, H, dtype=map.dtype, device=map.device).view(1, H, 1)
    fx = (map[:, 0] + 1) * (W - 1) / 2.0 - xx
    fy = (map[:, 1] + 1) * (H - 1) / 2.0 - yy
    flow = torch.stack((fx, fy), dim=1)
    if not output_channel_first:
        flow = flow.permute(0, 2, 3, 1)
    return flow


def convert_flow_to_mapping(flow, output_channel_first=True):
    """matching_utils.py (4-D case): flow in pixels -> absolute sampling position (x, y) of every pixel."""
    if flow.dim() != 4:
        raise RuntimeError("convert_flow_to_mapping: expects (B,2,H,W)")
    if flow.shape[1] != 2:
        flow = flow.permute(0, 3, 1, 2)
    B, C, H, W = flow.shape
    xx = torch.arange(0, W, dtype=flow.dtype, device=flow.device).view(1, 1, W)
    yy = torch.arange(0, H, dtype=flow.dtype, device=flow.device).view(1, H, 1)
    mapping = torch.stack((flow[:, 0] + xx, flow[:, 1] + yy), dim=1)
    return mapping if output_channel_first else mapping.permute(0, 2, 3, 1)


def get_gt_correspondence_mask(flow):
    """matching_utils.py:60-74: True where the flow points inside the image (borders included)."""
    m = convert_flow_to_mapping(flow)
    h, w = m.shape[-2:]
    return (m[:, 0] >= 0) & (m[:, 0] <= w - 1) & (m[:, 1] >= 0) & (m[:, 1] <= h - 1)


def estimate_probability_of_confidence_interval_of_mixture_density(uncert_output, R=1.0):
    """matching_utils.py:52-57 (Gaussian only)."""
    assert uncert_output.shape[1] == 1
    var = torch.exp(uncert_output)
    return 1.0 - torch.exp(-R ** 2 / (2 * var))


def align_tail(logits_ref, flow_q, logvar_q, return_flow=False):
    """Fused tail of align() (segmentation_model.py:514-522): bilinear-upsample the quarter-res flow / log-variance
    to the logits' size, confidence P_R, warp the reference logits.  Returns (warped, mask, cert[, flow_up])."""
    logits_ref = require_device_tensor(logits_ref.float().contiguous(), "logits_ref", torch.float32)
    flow_q = require_device_tensor(flow_q.float().contiguous(), "flow_q", torch.float32)
    logvar_q = require_device_tensor(logvar_q.float().contiguous(), "logvar_q", torch.float32)
    dev = same_device(logits_ref, flow_q, logvar_q)
    B, C, H, W = logits_ref.shape
    h, w = flow_q.shape[-2:]
    if tuple(flow_q.shape) != (B, 2, h, w) or tuple(logvar_q.shape) != (B, 1, h, w):
        raise RuntimeError("align_tail: flow_q must be (B,2,h,w) and logvar_q (B,1,h,w)")
    warped = torch.empty_like(logits_ref)
    mask = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    cert = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    flow_up = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev) if return_flow else None
    lib = _lib.load_library()
    with on_device(dev):
        rc = lib.rfn_align_tail_f32(ptr(logits_ref), ptr(flow_q), ptr(logvar_q), ptr(warped), ptr(mask), ptr(cert),
                                    ptr(flow_up), B, C, H, W, h, w, current_stream(dev))
    _lib.check(rc, "align_tail")
    if return_flow:
        return warped, mask.view(torch.bool), cert, flow_up
    return warped, mask.view(torch.bool), cert
