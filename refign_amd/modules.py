"""Host mirror of the correlation layers of models/modules.py (same class names and forward signatures), backed by the
HIP kernels.  The conv decoders that consume these volumes live in refign_amd/align.py."""
import torch
import torch.nn as nn

from . import _lib
from ._tensor import current_stream, ptr, require_device_tensor, same_device, on_device
from .correlation import local_correlation_layer, spatial_correlation_sample


class LocalFeatureCorrelationLayer(nn.Module):
    """models/modules.py:247-274.  forward(feature_source, feature_target) -> (B, 81, H, W), ReLU'd and L2-normalised.
    With grad enabled it composes the autograd sampler with torch ops (matcher training); under
    no_grad -- the UDA step (segmentation_model.py:194) -- it is one fused kernel."""

    def __init__(self, patch_size=9):
        super().__init__()
        self.local_correlation = spatial_correlation_sample
        self.patch_size = patch_size

    def forward(self, feature_source, feature_target, flow=None):
        needs_grad = torch.is_grad_enabled() and (feature_source.requires_grad or feature_target.requires_grad
                                                  or (flow is not None and flow.requires_grad))
        if self.patch_size == 9 and not needs_grad:
            return local_correlation_layer(feature_source.float().contiguous(), feature_target.float().contiguous(),
                                           None if flow is None else flow.float().contiguous())
        if flow is not None:
            from .matching import warp
            feature_source = warp(feature_source, flow)
        b = feature_target.shape[0]
        corr = self.local_correlation(feature_target, feature_source, patch_size=self.patch_size)
        corr = corr.view(b, self.patch_size * self.patch_size, feature_target.shape[2], feature_target.shape[3])
        return nn.functional.normalize(nn.functional.relu(corr), p=2, dim=1)


class GlobalFeatureCorrelationLayer(nn.Module):
    """models/modules.py:277-392 ('3D', H-first ordering of the source axis; mutual matching when
    cyclic_consistency).  forward(feature_source, feature_target) -> (B, Hs*Ws, Ht, Wt)."""

    def __init__(self, cyclic_consistency=True):
        super().__init__()
        self.cyclic_consistency = cyclic_consistency

    def forward(self, feature_source, feature_target):
        fs = require_device_tensor(feature_source.float().contiguous(), "feature_source", torch.float32)
        ft = require_device_tensor(feature_target.float().contiguous(), "feature_target", torch.float32)
        dev = same_device(fs, ft)
        B, C, hs, ws = fs.shape
        B2, C2, ht, wt = ft.shape
        if B != B2 or C != C2:
            raise RuntimeError("GlobalFeatureCorrelationLayer: batch/channel mismatch")
        out = torch.empty((B, hs * ws, ht, wt), dtype=torch.float32, device=dev)
        ws_ = torch.empty((B, hs * ws), dtype=torch.float32, device=dev) if self.cyclic_consistency else None
        lib = _lib.load_library()
        with on_device(dev):
            rc = lib.rfn_global_corr_layer_f32(ptr(fs), ptr(ft), ptr(out), None if ws_ is None else ptr(ws_), B, C, hs, ws, ht, wt,
                                               1 if self.cyclic_consistency else 0, current_stream(dev))
        _lib.check(rc, "GlobalFeatureCorrelationLayer")
        return out
