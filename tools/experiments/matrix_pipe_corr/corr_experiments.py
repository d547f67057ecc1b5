"""tools/experiments/matrix_pipe_corr/corr_experiments.py -- ctypes bindings of the matrix-pipe correlation EXPERIMENTS
(round 2: fp32 MFMA, corr_mfma.hip; round 3: split-fp16 MFMA, corr_f16.hip + split_f16.hip).  Both are exact and both are
slower than the shipped VALU kernel (profiles/r02_corr_mfma_*.txt, r03_corr_f16_ablation.txt); they were exported from the
product library until round 3 and live here since round 4:  make -C tools/experiments/matrix_pipe_corr  builds
lib/libcorr_experiments.so;  python -m pytest tools/experiments/matrix_pipe_corr -m gpu  runs their checks."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
from refign_amd._tensor import current_stream, on_device, ptr, require_device_tensor  # noqa: E402
from refign_amd.correlation import local_correlation_layer, spatial_correlation_sample  # noqa: E402,F401

_LIB = None


class _Shim:
    """the two entry points under the names the wrappers below were written with"""

    def __init__(self, lib):
        c_int, vp = ctypes.c_int, ctypes.c_void_p
        lib.rfx_split_f16.restype = c_int
        lib.rfx_split_f16.argtypes = [vp] * 3 + [c_int] * 4 + [vp]
        lib.rfn_local_corr_layer_f16split.restype = c_int
        lib.rfn_local_corr_layer_f16split.argtypes = [vp] * 3 + [c_int] * 5 + [vp]
        lib.rfx_corr9_mfma.restype = c_int
        lib.rfx_corr9_mfma.argtypes = [vp] * 3 + [c_int] * 5 + [vp]
        self.rfn_split_f16 = lib.rfx_split_f16
        self.rfn_local_corr_layer_f16split = lib.rfn_local_corr_layer_f16split
        self.rfx_corr9_mfma = lib.rfx_corr9_mfma


def load_library():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "lib", "libcorr_experiments.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: make -C {HERE}")
        _LIB = _Shim(ctypes.CDLL(path))
    return _LIB


class _check:
    @staticmethod
    def check(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: rc {rc}")


_lib = type("L", (), {"load_library": staticmethod(load_library), "check": staticmethod(_check.check)})


def split_f16(features, flow=None):
    """fp32 (B, C, H, W) features -> the split-fp16 chunk-major operand of `local_correlation_layer_split`
    ((B, C/32, 2, H, W, 32) fp16: hi = fp16(v), lo = fp16(v - hi)); with `flow` the features are bilinearly warped on the
    way (warp(feature_source, flow), uawarpc.py:149-152) without materialising the warped fp32 map."""
    require_device_tensor(features, "features", torch.float32)
    B, C, H, W = features.shape
    if C % 32:
        raise RuntimeError("split_f16: C must be a multiple of 32")
    if flow is not None:
        require_device_tensor(flow, "flow", torch.float32)
        if tuple(flow.shape) != (B, 2, H, W):
            raise RuntimeError("split_f16: flow must be (B,2,H,W)")
    out = torch.empty((B, C // 32, 2, H, W, 32), dtype=torch.float16, device=features.device)
    with on_device(features.device):
        rc = _lib.load_library().rfn_split_f16(ptr(features), ptr(flow), ptr(out), B, C, H, W,
                                               current_stream(features.device))
    _lib.check(rc, "split_f16")
    return out


def local_correlation_layer_split(source_split, target_split, fuse=True):
    """LocalFeatureCorrelationLayer.forward (modules.py:266-274) on the matrix pipe (csrc/corr_f16.hip) from operands made
    by `split_f16`: (B, 81, H, W) fp32 = patch-9 correlation (+ ReLU + L2 norm over the shifts when `fuse`)."""
    for t, name in ((source_split, "source_split"), (target_split, "target_split")):
        require_device_tensor(t, name, torch.float16)
    if source_split.shape != target_split.shape or source_split.dim() != 6 or source_split.shape[2] != 2 \
            or source_split.shape[5] != 32:
        raise RuntimeError("local_correlation_layer_split: operands must both be (B, C/32, 2, H, W, 32)")
    B, NC, _, H, W, _ = source_split.shape
    out = torch.empty((B, 81, H, W), dtype=torch.float32, device=source_split.device)
    with on_device(out.device):
        rc = _lib.load_library().rfn_local_corr_layer_f16split(ptr(target_split), ptr(source_split), ptr(out), B, NC * 32, H, W,
                                                               1 if fuse else 0, current_stream(out.device))
    _lib.check(rc, "local_corr_layer_f16split")
    return out



def corr9_mfma(in1, in2, fuse=False):
    """the fp32-matrix-pipe patch-9 forward (corr_mfma.hip; RFN_CORR_MFMA_CFG selects wave-private rings / shared tiles)"""
    for t, name in ((in1, "in1"), (in2, "in2")):
        require_device_tensor(t, name, torch.float32)
    B, C, H, W = in1.shape
    out = torch.empty((B, 81, H, W), dtype=torch.float32, device=in1.device)
    with on_device(out.device):
        rc = load_library().rfx_corr9_mfma(ptr(in1), ptr(in2), ptr(out), B, C, H, W, 1 if fuse else 0, current_stream(out.device))
    if rc > 0:
        return None                                                # shape not taken by the kernel
    _lib.check(rc, "corr9_mfma")
    return out
