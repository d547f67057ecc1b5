"""Spatial correlation sampler -- host side of the drop-in boundary.

Mirrors, name for name, what the reference exposes for this op:
  * `forward(input1, input2, kH, kW, patchH, patchW, padH, padW, dilationH, dilationW, dilation_patchH,
    dilation_patchW, dH, dW)` and `backward(input1, input2, grad_output, <same 12 ints>)` -- the two functions of the
    pybind module `models.correlation_ops.correlation` (correlation_sampler.cpp:62-70,92-101,129-132);
  * `spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
    dilation_patch=1)` and `SpatialCorrelationSamplerFunction` (correlation_function.py:14-94): fp32 under autocast,
    once-differentiable, saves both inputs.
Inputs are borrowed and must be contiguous NCHW on one HIP device; outputs are freshly allocated and returned by
value (the reference: torch::zeros / zeros_like, correlation_cuda_kernel.cu:259,291-292).  Errors are RuntimeError.
"""
import os

import torch
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import _lib
from ._tensor import current_stream, ptr, require_device_tensor, same_device, on_device

# (half: the pybind module of the reference dispatches it on the device, correlation_cuda_kernel.cu:267; `forward` / `backward`
# below take it, `spatial_correlation_sample` casts to float32 like the reference's wrapper and never passes it)
_SUFFIX = {torch.float32: "f32", torch.float64: "f64", torch.float16: "f16"}


def _suffix(t):
    try:
        return _SUFFIX[t.dtype]
    except KeyError:
        raise RuntimeError(f"correlation: unsupported dtype {t.dtype} (float32 / float64 / float16)") from None


def output_size(iH, iW, kH, kW, padH, padW, dilationH, dilationW, dH, dW):
    """correlation.cpp:94-99."""
    oH = (iH + 2 * padH - ((kH - 1) * dilationH + 1)) // dH + 1
    oW = (iW + 2 * padW - ((kW - 1) * dilationW + 1)) // dW + 1
    return oH, oW


def forward(input1, input2, kH, kW, patchH, patchW, padH, padW, dilationH, dilationW, dilation_patchH,
            dilation_patchW, dH, dW):
    require_device_tensor(input1, "input1")
    require_device_tensor(input2, "input2", input1.dtype)
    dev = same_device(input1, input2)
    if input1.dim() != 4 or input1.shape != input2.shape:
        raise RuntimeError("correlation.forward: input1/input2 must both be (B,C,H,W) of the same shape")
    B, C, iH, iW = input1.shape
    oH, oW = output_size(iH, iW, kH, kW, padH, padW, dilationH, dilationW, dH, dW)
    if oH <= 0 or oW <= 0:
        raise RuntimeError(f"correlation.forward: empty output {oH}x{oW}")
    out = torch.empty((B, patchH, patchW, oH, oW), dtype=input1.dtype, device=dev)
    lib = _lib.load_library()
    fn = getattr(lib, "rfn_corr_fwd_" + _suffix(input1))
    with on_device(dev):
        rc = fn(ptr(input1), ptr(input2), ptr(out), B, C, iH, iW, kH, kW, patchH, patchW, padH, padW, dilationH,
                dilationW, dilation_patchH, dilation_patchW, dH, dW, current_stream(dev))
    _lib.check(rc, "correlation.forward")
    return out


def backward(input1, input2, grad_output, kH, kW, patchH, patchW, padH, padW, dilationH, dilationW,
             dilation_patchH, dilation_patchW, dH, dW):
    require_device_tensor(input1, "input1")
    require_device_tensor(input2, "input2", input1.dtype)
    grad_output = grad_output.contiguous()
    require_device_tensor(grad_output, "grad_output", input1.dtype)
    dev = same_device(input1, input2, grad_output)
    B, C, iH, iW = input1.shape
    oH, oW = output_size(iH, iW, kH, kW, padH, padW, dilationH, dilationW, dH, dW)
    if tuple(grad_output.shape) != (B, patchH, patchW, oH, oW):
        raise RuntimeError(f"correlation.backward: grad_output shape {tuple(grad_output.shape)} != "
                           f"{(B, patchH, patchW, oH, oW)}")
    g1 = torch.empty_like(input1)
    g2 = torch.empty_like(input2)
    lib = _lib.load_library()
    fn = getattr(lib, "rfn_corr_bwd_" + _suffix(input1))
    with on_device(dev):
        rc = fn(ptr(input1), ptr(input2), ptr(grad_output), ptr(g1), ptr(g2), B, C, iH, iW, kH, kW, patchH, patchW,
                padH, padW, dilationH, dilationW, dilation_patchH, dilation_patchW, dH, dW, current_stream(dev))
    _lib.check(rc, "correlation.backward")
    return [g1, g2]


class SpatialCorrelationSamplerFunction(torch.autograd.Function):
    """correlation_function.py:46-94."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                dilation_patch=1):
        input1 = input1.contiguous()
        input2 = input2.contiguous()
        ctx.save_for_backward(input1, input2)
        ctx.geometry = (*_pair(kernel_size), *_pair(patch_size), *_pair(padding), *_pair(dilation),
                        *_pair(dilation_patch), *_pair(stride))
        return forward(input1, input2, *ctx.geometry)

    @staticmethod
    @once_differentiable
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        g1, g2 = backward(input1, input2, grad_output, *ctx.geometry)
        return g1, g2, None, None, None, None, None, None


def spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                               dilation_patch=1):
    """Same signature and semantics as correlation_function.py:14-43.  Every parameter except the inputs may be an
    int or a pair.  Returns (B, patchH, patchW, oH, oW)."""
    return SpatialCorrelationSamplerFunction.apply(input1, input2, kernel_size, patch_size, stride, padding,
                                                   dilation, dilation_patch)


def local_correlation_layer(feature_source, feature_target, flow=None, single_kernel_warp=False):
    """LocalFeatureCorrelationLayer.forward (modules.py:266-274) as ONE kernel: patch-9 correlation (input1 = target,
    input2 = source) + ReLU + L2-normalisation over the 81 shifts -> (B,81,H,W).

    With `flow` (B,2,H,W; pixels of this level) the source features are first bilinearly warped
    (warp(feature_source, flow) of uawarpc.py:149-152).  Two implementations, same numbers:
      * default: warp kernel into a scratch map (L2/Infinity-Cache resident at these sizes), then the LDS-DMA
        correlation kernel -- measured faster (K4 level 1: 233 + 145 us) than
      * single_kernel_warp=True: the register-staged kernel that warps while it stages the source tile and never
        materialises the warped map (796 us: the per-element gathers cannot use the LDS-DMA path).
    Inference-only (the UDA step runs align under no_grad, segmentation_model.py:194).
    """
    require_device_tensor(feature_target, "feature_target", torch.float32)
    require_device_tensor(feature_source, "feature_source", torch.float32)
    if flow is not None:
        require_device_tensor(flow, "flow", torch.float32)
    dev = same_device(feature_target, feature_source, flow)
    if feature_target.shape != feature_source.shape or feature_target.dim() != 4:
        raise RuntimeError("local_correlation_layer: features must both be (B,C,H,W) of the same shape")
    B, C, H, W = feature_target.shape
    if flow is not None and tuple(flow.shape) != (B, 2, H, W):
        raise RuntimeError("local_correlation_layer: flow must be (B,2,H,W)")
    if flow is not None and not single_kernel_warp:
        from .matching import warp_nocheck
        feature_source, flow = warp_nocheck(feature_source, flow), None
    out = torch.empty((B, 81, H, W), dtype=torch.float32, device=dev)
    lib = _lib.load_library()
    splits = _channel_splits(B, C, H, W) if flow is None else 1
    with on_device(dev):
        if splits > 1:
            ws = _split_workspace(lib.rfn_local_corr_layer_split_workspace_bytes(B, H, W, splits), dev)
            rc = lib.rfn_local_corr_layer_split_f32(ptr(feature_target), ptr(feature_source), ptr(out), ptr(ws), B, C, H, W,
                                                    splits, current_stream(dev))
        else:
            rc = lib.rfn_local_corr_layer_f32(ptr(feature_target), ptr(feature_source), ptr(flow), ptr(out), B, C, H, W,
                                              current_stream(dev))
    _lib.check(rc, "local_correlation_layer")
    return out


_SPLIT_WS = {}


def _split_workspace(nbytes, dev):
    """Workspace of the channel-split path: its tail holds the tickets of the one-launch form, which must be zero before the first
    call and are left zero by every call (include/refign_hip.h).  Eager calls: one per (device, stream, size), kept -- two
    streams may run the same level side by side.  Captured calls: one per call (below)."""
    if torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture the current stream is torch's capture stream whichever stream will replay the graph: a cached
        # workspace would be shared by every captured graph, and two graphs replaying side by side would race on its tickets and
        # partial sums.  One workspace per captured call, from the graph's own pool, zeroed by a fill KERNEL node.
        return torch.empty(nbytes, dtype=torch.uint8, device=dev).fill_(0)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream, nbytes)
    ws = _SPLIT_WS.get(key)
    if ws is None:
        # never evicted: a captured hipGraph may hold this address (a few MB per distinct (stream, level size))
        ws = _SPLIT_WS[key] = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    return ws


def _channel_splits(B, C, H, W):
    """How many channel chunks the patch-9 kernel of a TINY map is split into (csrc/corr.hip launch_corr9_split): the
    tiled kernel walks the channels of a tile serially, so a map of a few 8x64 tiles (2 x 256 x 32 x 32: 8 workgroups on
    256 CUs) is ~100 us of pure latency.  Measured (profiles/r02_kbench_corr_split.txt): it pays for such maps only --
    at 2 x 256 x 135 x 240 (136 tiles) two or four chunks are SLOWER than the one-kernel path (127 / 120 vs 107 us: the
    per-workgroup set-up and the extra pass over the partial sums cost more than the second half of the chip gives).
    1 = the one-kernel path; RFN_CORR_SPLIT=0 switches the split off, RFN_CORR_SPLITS=n forces n chunks."""
    if os.environ.get("RFN_CORR_SPLIT", "1") == "0" or W % 4 or C % 8:
        return 1
    forced = int(os.environ.get("RFN_CORR_SPLITS", "0"))                # tuning knob
    if forced:
        return forced if (forced > 1 and C % forced == 0 and (C // forced) % 8 == 0) else 1
    tiles = B * -(-W // 32) * -(-H // 8)
    if tiles > 64:
        return 1
    # chunks of >= 64 channels in multiples of 32 take the one-launch form (four wave groups per workgroup, joined by the
    # tile's last workgroup): split down to 64 channels; where that gives no split, the two-launch form down to 16
    def grow(floor_c, mult):
        s = 1
        while tiles * s < 64 and s < 8 and C % (2 * s) == 0 and (C // (2 * s)) % mult == 0 and C // (2 * s) >= floor_c:
            s *= 2
        return s
    s = grow(64, 32)
    return s if s > 1 else grow(16, 8)
