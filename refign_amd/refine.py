"""Host mirror of DomainAdaptationSegmentationModel.refine / .eta (models/segmentation_model.py:438-491) and the
pseudo-label part of get_dacs_mix (segmentation_model.py:551-556), backed by csrc/refine.hip."""
import torch

from . import _lib
from ._tensor import current_stream, ptr, require_device_tensor, same_device, on_device


@torch.no_grad()
def refine(logits_trg, logits_ref, warp_mask, certs, gamma=0.25, disable_M=False, disable_P=False):
    """Adaptive label correction.  logits_*: (B,19,H,W); warp_mask: (B,H,W) bool or None; certs: (B,1,H,W) or None.
    Returns the refined target probabilities (B,19,H,W) -- NOT renormalised, exactly like the reference."""
    logits_trg = require_device_tensor(logits_trg.float().contiguous(), "logits_trg", torch.float32)
    logits_ref = require_device_tensor(logits_ref.float().contiguous(), "logits_ref", torch.float32)
    dev = same_device(logits_trg, logits_ref, warp_mask, certs)
    B, C, H, W = logits_trg.shape
    assert C == 19, 'we assume cityscapes classes'   # segmentation_model.py:441
    if logits_ref.shape != logits_trg.shape:
        raise RuntimeError("refine: logits_trg / logits_ref shape mismatch")
    m8 = None
    if warp_mask is not None:
        if tuple(warp_mask.shape) != (B, H, W):
            raise RuntimeError("refine: warp_mask must be (B,H,W)")
        m8 = warp_mask.contiguous()
        m8 = m8.view(torch.uint8) if m8.dtype == torch.bool else m8.to(torch.uint8)
    if certs is not None:
        certs = require_device_tensor(certs.float().contiguous(), "certs", torch.float32)
        if certs.numel() != B * H * W:
            raise RuntimeError("refine: certs must be (B,1,H,W)")
    lib = _lib.load_library()
    ws = torch.empty(lib.rfn_refine_workspace_bytes(B), dtype=torch.uint8, device=dev)
    out = torch.empty_like(logits_trg)
    flags = (1 if disable_M else 0) | (2 if disable_P else 0)
    with on_device(dev):
        rc = lib.rfn_refine_f32(ptr(logits_trg), ptr(logits_ref), ptr(m8), ptr(certs), ptr(out), ptr(ws), B, C, H, W,
                                float(gamma), flags, current_stream(dev))
    _lib.check(rc, "refine")
    return out
