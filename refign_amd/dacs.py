"""N4 -- GPU-side data step of the UDA iteration: DACS class-mix + colour jitter + Gaussian blur as HIP kernels
(csrc/dacs.hip) behind the reference's `get_dacs_mix` (models/segmentation_model.py:525-582) /
`strong_transform` (helpers/dacs_transforms.py:14-24).

The random DECISIONS stay on the host and are drawn exactly where the reference draws them (python `random` for the two
coins, numpy for the class choice and the blur sigma, torch's CPU generator for the jitter -- kornia's ColorJitter order:
a permutation of the four operators, one uniform per operator, one more for the hue angle); what moves to the device is
the pixel work: per step 2 x (3 + 1 + 1) full-resolution maps mixed, jittered and blurred in 2-4 launches instead of ~60
element-wise / reduction / grouped-convolution launches.  The class set of a sample reaches the kernel as a DEVICE bit set
built from torch.unique's device result, so the step keeps running without a host synchronisation.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from ._tensor import current_stream, on_device, ptr, upload_async

IMNET_MEAN = (0.485, 0.456, 0.406)
IMNET_STD = (0.229, 0.224, 0.225)
_YIQ = np.array([[0.299, 0.587, 0.114], [0.596, -0.274, -0.322], [0.211, -0.523, 0.312]])
_YIQ_INV = np.linalg.inv(_YIQ)
MAX_BATCH = 8


def usable(images_src, images_trg, gt_src, num_classes=19):
    """num_classes: labels are 0 .. num_classes - 1 and 255.  The kernels carry a sample's chosen classes as a 32-bit set
    (bit c = class c for c <= 30, bit 31 = the ignore label 255): label sets with more than 31 classes take the torch path."""
    return (num_classes <= 31 and images_src.is_cuda and images_src.dtype == torch.float32 and images_trg.dtype == torch.float32
            and images_src.dim() == 4 and images_src.shape[1] == 3 and images_trg.shape == images_src.shape
            and images_src.shape[0] <= MAX_BATCH and (images_src.shape[2] * images_src.shape[3]) % 4 == 0
            and min(images_src.shape[2:]) > 16 and gt_src.dtype == torch.long)


def draw_class_bits(classes, nb):
    """get_class_masks' draws (dacs_transforms.py:81-92: numpy choice of half of the batch-wide class set per sample) ->
    (nb,) int64 DEVICE bit sets: bit c = class c, bit 31 = the ignore label 255."""
    n = classes.shape[0]
    idx = np.stack([np.random.choice(n, int((n + n % 2) / 2), replace=False) for _ in range(nb)])
    chosen = classes[upload_async(idx, torch.long, classes.device)]                  # (nb, k) class values
    chosen = torch.where(chosen == 255, torch.full_like(chosen, 31), chosen).clamp_(0, 31)
    bits = torch.bitwise_left_shift(torch.ones_like(chosen), chosen)
    out = bits[:, 0]
    for j in range(1, bits.shape[1]):                   # OR, not sum: a repeated (clamped) class must not carry into the next bit
        out = torch.bitwise_or(out, bits[:, j])
    return out


def draw_jitter(s):
    """One sample's colour-jitter draws in the order of uda._color_jitter (kornia's ColorJitter.generate_parameters):
    (order[4], factor[4], hue 3x3)."""
    order = torch.randperm(4).tolist()
    factor = [1.0] * 4
    hue = np.eye(3)
    lo = max(0.0, 1 - s)
    for op in order:
        u = float(torch.rand(()))
        factor[op] = lo + u * (1 + s - lo)
        if op == 3:
            h = (2.0 * float(torch.rand(())) - 1.0) * s * 2 * math.pi
            c, sn = math.cos(h), math.sin(h)
            hue = _YIQ_INV @ np.array([[1, 0, 0], [0, c, -sn], [0, sn, c]]) @ _YIQ
    return order, factor, hue


def mix(images_src, images_trg, gt_src, pseudo_label, pseudo_weight, class_bits, jitter, blur_sigma, part="all"):
    """jitter: per sample None or (order, factor, hue); blur_sigma: per sample None or sigma.  Returns
    (mixed_img (B,3,H,W) fp32, mixed_lbl (B,H,W) int64, mixed_weight (B,H,W) fp32).
    `part`: "all", or one half of the mix -- "image" (mixed_img only; pseudo_label / pseudo_weight may be None: the mixed
    image is a function of the two images and the source labels alone, so the student's forward on it need not wait for the
    teacher) / "labels" (mixed_lbl, mixed_weight only).  Both halves take the class set from `class_bits`."""
    assert part in ("all", "image", "labels")
    B, H, W = gt_src.shape
    dev = gt_src.device
    gt = gt_src.contiguous()
    img = lbl = wgt = src = trg = ps = pw = None
    if part != "labels":
        src, trg = images_src.contiguous(), images_trg.contiguous()
        img = torch.empty_like(src)
    if part != "image":
        ps = pseudo_label.contiguous()
        pw = pseudo_weight.to(torch.float32).contiguous()
        lbl = torch.empty_like(gt)
        wgt = torch.empty_like(pw)
    ws = torch.empty(MAX_BATCH, dtype=torch.float64, device=dev)
    on = (ctypes.c_int * B)(*[0 if j is None else 1 for j in jitter])
    order = (ctypes.c_int * (4 * B))(*[v for j in jitter for v in ([0, 1, 2, 3] if j is None else j[0])])
    factor = (ctypes.c_float * (4 * B))(*[v for j in jitter for v in ([1.0] * 4 if j is None else j[1])])
    hue = (ctypes.c_float * (9 * B))(*[float(v) for j in jitter
                                       for v in (np.eye(3) if j is None else np.asarray(j[2])).reshape(-1)])
    mean3, std3 = (ctypes.c_float * 3)(*IMNET_MEAN), (ctypes.c_float * 3)(*IMNET_STD)
    lib = _lib.load_library()
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)  # noqa: E731
    optr = lambda t: None if t is None else ptr(t)  # noqa: E731
    with on_device(dev):
        rc = lib.rfn_dacs_mix_jitter(optr(src), optr(trg), ptr(gt), optr(ps), optr(pw), optr(img), optr(lbl), optr(wgt), ptr(ws),
                                     B, H, W, ptr(class_bits.contiguous()), cast(on), cast(order), cast(factor), cast(hue),
                                     cast(mean3), cast(std3), current_stream(dev))
        _lib.check(rc, "dacs_mix_jitter")
        if img is not None and any(s is not None for s in blur_sigma):
            bon = (ctypes.c_int * B)(*[0 if s is None else 1 for s in blur_sigma])
            sig = (ctypes.c_double * B)(*[1.0 if s is None else float(s) for s in blur_sigma])
            tmp, out = torch.empty_like(img), torch.empty_like(img)
            # kornia's window (dacs_transforms.py:68-72): ~0.1 x the extent, odd
            ks = [int(np.floor(np.ceil(0.1 * d) - 0.5 + np.ceil(0.1 * d) % 2)) for d in (H, W)]
            rc = lib.rfn_dacs_blur(ptr(img), ptr(tmp), ptr(out), B, 3, H, W, ks[0], ks[1], cast(bon), cast(sig), cast(sig),
                                   current_stream(dev))
            _lib.check(rc, "dacs_blur")
            img = out
    return img, lbl, wgt
