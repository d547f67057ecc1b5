"""oracle/cpu_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy + torch-CPU ATen, float32/float64) of the reference's align-and-refine
hot path, function by function.  Each function cites the reference file:line it follows
(paths relative to /root/reference).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; nothing in refign_amd/ does.

Parity pin: every function here is checked in tests/test_oracle_cpu.py against golden vectors in
tests/golden/*.npz that were produced by importing the reference itself in the authoring
container (tests/golden/make_golden.py).  The reference ships no tests of its own,
so those vectors are the pin.

Third-party arithmetic: the reference's own path is built on torch ATen ops (conv2d, grid_sample,
interpolate, softmax ...; requirements.txt:1 pins torch==1.7.1, here 2.10).  Where a function below
is a thin composition of ATen ops the restatement calls the same ATen op on CPU; where the
reference calls its own native code (correlation sampler) or where we replace an ATen op by a HIP
kernel (grid_sample warp, softmax/entropy/blend of refine, bmm of the global correlation) the
restatement is explicit numpy so the HIP kernel is checked against independent arithmetic.
"""
import ctypes
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = ctypes.CDLL(path)
    return _LIB


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


# ----------------------------------------------------------------------------------------------
# a1-a3: spatial correlation sampler  (models/correlation_ops/correlation.cpp:80-183)
# ----------------------------------------------------------------------------------------------
def corr_out_size(iH, iW, kernel_size=1, padding=0, dilation=1, stride=1):
    kH, kW = _pair(kernel_size)
    pH, pW = _pair(padding)
    dlH, dlW = _pair(dilation)
    sH, sW = _pair(stride)
    oH = (iH + 2 * pH - ((kH - 1) * dlH + 1)) // sH + 1   # correlation.cpp:98
    oW = (iW + 2 * pW - ((kW - 1) * dlW + 1)) // sW + 1   # correlation.cpp:99
    return oH, oW


def corr_forward(in1, in2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1, dilation_patch=1):
    """correlation_cpp_forward (correlation.cpp:80-129) via oracle/corr_oracle.c."""
    in1 = np.ascontiguousarray(in1)
    in2 = np.ascontiguousarray(in2)
    assert in1.dtype == in2.dtype and in1.dtype in (np.float32, np.float64)
    B, C, iH, iW = in1.shape
    kH, kW = _pair(kernel_size)
    pH, pW = _pair(patch_size)
    padH, padW = _pair(padding)
    dlH, dlW = _pair(dilation)
    dpH, dpW = _pair(dilation_patch)
    sH, sW = _pair(stride)
    oH, oW = corr_out_size(iH, iW, (kH, kW), (padH, padW), (dlH, dlW), (sH, sW))
    out = np.empty((B, pH, pW, oH, oW), dtype=in1.dtype)
    fn = getattr(_lib(), "oracle_corr_fwd_f32" if in1.dtype == np.float32 else "oracle_corr_fwd_f64")
    fn.restype = None
    fn(in1.ctypes.data_as(ctypes.c_void_p), in2.ctypes.data_as(ctypes.c_void_p),
       out.ctypes.data_as(ctypes.c_void_p), *[ctypes.c_int(v) for v in
                                              (B, C, iH, iW, kH, kW, pH, pW, padH, padW, dlH, dlW, dpH, dpW, sH, sW)])
    return out


def corr_backward(in1, in2, gout, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                  dilation_patch=1):
    """correlation_cpp_backward (correlation.cpp:131-183) via oracle/corr_oracle.c."""
    in1 = np.ascontiguousarray(in1)
    in2 = np.ascontiguousarray(in2)
    gout = np.ascontiguousarray(gout)
    B, C, iH, iW = in1.shape
    kH, kW = _pair(kernel_size)
    pH, pW = _pair(patch_size)
    padH, padW = _pair(padding)
    dlH, dlW = _pair(dilation)
    dpH, dpW = _pair(dilation_patch)
    sH, sW = _pair(stride)
    oH, oW = gout.shape[3], gout.shape[4]
    g1 = np.empty_like(in1)
    g2 = np.empty_like(in2)
    fn = getattr(_lib(), "oracle_corr_bwd_f32" if in1.dtype == np.float32 else "oracle_corr_bwd_f64")
    fn.restype = None
    fn(in1.ctypes.data_as(ctypes.c_void_p), in2.ctypes.data_as(ctypes.c_void_p),
       gout.ctypes.data_as(ctypes.c_void_p), g1.ctypes.data_as(ctypes.c_void_p),
       g2.ctypes.data_as(ctypes.c_void_p), *[ctypes.c_int(v) for v in
                                             (B, C, iH, iW, oH, oW, kH, kW, pH, pW, padH, padW, dlH, dlW, dpH, dpW,
                                              sH, sW)])
    return g1, g2


def l2_normalize(x, axis=1, eps=1e-12):
    """torch.nn.functional.normalize(p=2): x / max(||x||_2, eps)."""
    n = np.sqrt(np.sum(x.astype(np.float64) ** 2, axis=axis, keepdims=True)).astype(x.dtype)
    return x / np.maximum(n, eps)


def local_correlation_layer(feature_source, feature_target, patch_size=9):
    """LocalFeatureCorrelationLayer.forward (models/modules.py:266-274).

    NB argument order (modules.py:268): input1 = TARGET feats, input2 = (warped) SOURCE feats.
    """
    corr = corr_forward(feature_target, feature_source, patch_size=patch_size)
    b, _, _, h, w = corr.shape
    corr = corr.reshape(b, patch_size * patch_size, h, w)
    return l2_normalize(np.maximum(corr, 0), axis=1)


# ----------------------------------------------------------------------------------------------
# a7: global correlation + mutual matching  (models/modules.py:294-375)
# ----------------------------------------------------------------------------------------------
def global_correlation_layer(feature_source, feature_target, cyclic_consistency=True):
    """GlobalFeatureCorrelationLayer.forward (modules.py:294-308), '3D' H-first branch
    (modules.py:361-375) + mutual_matching (modules.py:310-333).

    corr[b, hs*Ws+ws, ht, wt] = <f_src[b,:,hs,ws], f_trg[b,:,ht,wt]>
    """
    b, c, hs, ws = feature_source.shape
    _, _, ht, wt = feature_target.shape
    fs = feature_source.reshape(b, c, hs * ws)
    ft = feature_target.reshape(b, c, ht * wt)
    # (b, S, T): fp32 accumulate in fp64 then round -> independent of summation order
    corr = np.einsum('bcs,bct->bst', fs.astype(np.float64), ft.astype(np.float64)).astype(feature_source.dtype)
    if cyclic_consistency:
        eps = np.asarray(1e-5, dtype=corr.dtype)
        max_over_src = corr.max(axis=1, keepdims=True)        # corr4d_B_max (modules.py:320)
        max_over_trg = corr.max(axis=2, keepdims=True)        # corr4d_A_max (modules.py:321)
        cb = corr / (max_over_src + eps)
        ca = corr / (max_over_trg + eps)
        corr = corr * (ca * cb)                                # modules.py:331
    corr = corr.reshape(b, hs * ws, ht, wt)
    return l2_normalize(np.maximum(corr, 0), axis=1)


# ----------------------------------------------------------------------------------------------
# a12: warp (helpers/matching_utils.py:11-49) -- grid_sample restated explicitly
# ----------------------------------------------------------------------------------------------
def warp(x, flo, return_mask=False):
    """warp(): bilinear, align_corners=True, zero padding, strict-inequality validity mask,
    early return when the flow is identically zero (matching_utils.py:19-22)."""
    B, C, H, W = x.shape
    f32 = np.float32
    if np.all(flo == 0):
        if return_mask:
            return x, np.ones((B, H, W), dtype=bool)
        return x
    xx = np.arange(W, dtype=f32)[None, None, :]
    yy = np.arange(H, dtype=f32)[None, :, None]
    # matching_utils.py:35-36 (same op order, fp32)
    vx = (f32(2.0) * (xx + flo[:, 0].astype(f32))) / f32(max(W - 1, 1)) - f32(1.0)
    vy = (f32(2.0) * (yy + flo[:, 1].astype(f32))) / f32(max(H - 1, 1)) - f32(1.0)
    # ATen grid_sampler unnormalize, align_corners=True: ((coord + 1) / 2) * (size - 1)
    ix = ((vx + f32(1.0)) / f32(2.0)) * f32(W - 1)
    iy = ((vy + f32(1.0)) / f32(2.0)) * f32(H - 1)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    wx1 = (ix - x0).astype(f32)
    wy1 = (iy - y0).astype(f32)
    wx0 = f32(1.0) - wx1
    wy0 = f32(1.0) - wy1
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    out = np.zeros((B, C, H, W), dtype=f32)
    bidx = np.arange(B)[:, None, None]
    for dy, wy in ((0, wy0), (1, wy1)):
        for dx, wx in ((0, wx0), (1, wx1)):
            xs = x0 + dx
            ys = y0 + dy
            ok = (xs >= 0) & (xs < W) & (ys >= 0) & (ys < H)
            xs_c = np.clip(xs, 0, W - 1)
            ys_c = np.clip(ys, 0, H - 1)
            vals = x.astype(f32)[bidx, :, ys_c, xs_c]           # (B,H,W,C)
            wgt = (wx * wy * ok).astype(f32)
            out += np.moveaxis(vals, -1, 1) * wgt[:, None]
    if return_mask:
        mask = (vx > -1) & (vy > -1) & (vx < 1) & (vy < 1)       # matching_utils.py:46-47
        return out, mask
    return out


def unnormalise_and_convert_mapping_to_flow(mapping):
    """helpers/matching_utils.py:77-103 (4-D, channel-first branch)."""
    B, C, H, W = mapping.shape
    out = np.empty_like(mapping)
    xx = np.arange(W, dtype=mapping.dtype)[None, None, :]
    yy = np.arange(H, dtype=mapping.dtype)[None, :, None]
    out[:, 0] = (mapping[:, 0] + 1) * (W - 1) / 2.0 - xx
    out[:, 1] = (mapping[:, 1] + 1) * (H - 1) / 2.0 - yy
    return out


def confidence_from_logvar(log_var, R=1.0):
    """estimate_probability_of_confidence_interval_of_mixture_density (matching_utils.py:52-57)."""
    var = np.exp(log_var)
    return 1.0 - np.exp(-R ** 2 / (2 * var))


# ----------------------------------------------------------------------------------------------
# a17: refine + eta  (models/segmentation_model.py:438-491)
# ----------------------------------------------------------------------------------------------
def _softmax(x, axis=1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def _log_softmax(x, axis=1):
    m = x.max(axis=axis, keepdims=True)
    s = x - m
    return s - np.log(np.exp(s).sum(axis=axis, keepdims=True))


def eta(logits):
    """normalised entropy (segmentation_model.py:484-491)."""
    dim = logits.shape[1]
    ent = -(_softmax(logits) * _log_softmax(logits)).sum(axis=1)
    return ent / math.log(dim)


def refine(logits_trg, logits_ref, warp_mask, certs, gamma=0.25, disable_M=False, disable_P=False):
    """DomainAdaptationSegmentationModel.refine (segmentation_model.py:438-482).

    NB: epsilon is per channel, the output is NOT a simplex.
    """
    dt = logits_trg.dtype
    b, c, h, w = logits_trg.shape
    assert c == 19
    probs_trg = _softmax(logits_trg)
    probs_ref = _softmax(logits_ref)
    pred_trg = probs_trg.argmax(axis=1)
    pred_ref = probs_ref.argmax(axis=1)
    s = eta(logits_trg).astype(np.float64).mean(axis=(1, 2)).astype(dt) ** dt.type(gamma)   # :449
    static_large = np.array([0, 1, 2, 3, 4, 8, 9, 10])                                         # :452
    M = np.isin(pred_trg, static_large) & np.isin(pred_ref, static_large)                     # :453-458
    M = np.broadcast_to(M[:, None], probs_trg.shape).copy()
    M[:, 5:8] = False                                                                          # :460
    M[:, 11:] = False                                                                          # :461
    if disable_M:
        M[:] = False
    if disable_P:
        certs = None
    if certs is not None:
        P = np.broadcast_to(certs, probs_trg.shape)
    else:
        P = np.full_like(probs_trg, 0.5)
    eps = s.reshape(-1, 1, 1, 1) * np.maximum(P, M.astype(dt))                                 # :475
    if warp_mask is not None:
        eps = eps * warp_mask[:, None].astype(dt)                                              # :477-479
    return ((1 - eps) * probs_trg + eps * probs_ref).astype(dt)                                # :481


def pseudo_label(probs, threshold=0.968):
    """pseudo-label part of get_dacs_mix (segmentation_model.py:551-556): max/argmax over classes
    and the BATCH-GLOBAL confident-pixel fraction."""
    prob = probs.max(axis=1)
    label = probs.argmax(axis=1)
    weight = np.float32((prob >= threshold).sum() / prob.size)
    return prob, label, weight
