"""Multi-resolution fusion front end of the decode heads on the HIP kernel of csrc/upcat.hip.

`upsample_concat(token_maps, sizes, out_size)`: the per-stage embeddings -- token maps (n, h_l*w_l, C_l) straight from
the embedding Linear -- bilinearly up-sampled (align_corners=False) to `out_size` and concatenated along channels in ONE
pass, returned as an NCHW-shaped, channels-last tensor (n, sum C_l, H, W): what
`torch.cat([F.interpolate(t.transpose(1, 2).reshape(n, C_l, h_l, w_l), out_size, mode='bilinear') ...], 1)` gives
(daformer.py:205-222, segformer.py:86-104).  Backward: one gather kernel over the concatenated gradient for all levels
(rfn_upsample_concat_nhwc_bwd; RFN_UPCAT_BWD=0: the library's bilinear backward on the channel slices).
"""
import ctypes
import os

import torch

from . import _lib
from ._tensor import current_stream, on_device, ptr

_DT = {torch.float32: 0, torch.bfloat16: 1}


class _UpCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out_size, sizes, *maps):
        n = maps[0].shape[0]
        H, W = out_size
        cs = [m.shape[2] for m in maps]
        out = torch.empty((n, H, W, sum(cs)), dtype=maps[0].dtype, device=maps[0].device)
        arr = ctypes.c_int * len(maps)
        srcs = [ptr(m) for m in maps] + [None] * (4 - len(maps))
        lib = _lib.load_library()
        with on_device(out.device):
            rc = lib.rfn_upsample_concat_nhwc(srcs[0], srcs[1], srcs[2], srcs[3], arr(*[s[0] for s in sizes]),
                                              arr(*[s[1] for s in sizes]), arr(*cs), len(maps), ptr(out), n, H, W,
                                              _DT[out.dtype], current_stream(out.device))
        _lib.check(rc, "upsample_concat_nhwc")
        ctx.geom = (n, H, W, cs, sizes)
        return out.permute(0, 3, 1, 2)                                  # NCHW-shaped view, channels-last strides

    @staticmethod
    def backward(ctx, g):
        n, H, W, cs, sizes = ctx.geom
        if os.environ.get("RFN_UPCAT_BWD", "1") != "0" and g.dtype in _DT:
            gn = g.permute(0, 2, 3, 1)
            if not gn.is_contiguous():
                gn = gn.contiguous()
            grads = [torch.empty((n, h * w, c), dtype=g.dtype, device=g.device) for c, (h, w) in zip(cs, sizes)]
            arr = ctypes.c_int * len(cs)
            dsts = [ptr(t) for t in grads] + [None] * (4 - len(cs))
            with on_device(g.device):
                rc = _lib.load_library().rfn_upsample_concat_nhwc_bwd(
                    ptr(gn), dsts[0], dsts[1], dsts[2], dsts[3], arr(*[s_[0] for s_ in sizes]), arr(*[s_[1] for s_ in sizes]),
                    arr(*cs), len(cs), n, H, W, _DT[g.dtype], current_stream(g.device))
            _lib.check(rc, "upsample_concat_nhwc_bwd")
            return (None, None) + tuple(t if ctx.needs_input_grad[2 + i] else None for i, t in enumerate(grads))
        grads, off = [], 0
        for i, (c, (h, w)) in enumerate(zip(cs, sizes)):
            if not ctx.needs_input_grad[2 + i]:
                grads.append(None)
            else:
                gl = g[:, off:off + c]
                if (h, w) != (H, W):
                    gl = torch.ops.aten.upsample_bilinear2d_backward(gl, [H, W], [n, c, h, w], False, None, None)
                grads.append(gl.permute(0, 2, 3, 1).reshape(n, h * w, c))
            off += c
        return (None, None) + tuple(grads)


def upsample_concat(token_maps, sizes, out_size):
    """token_maps[l]: (n, h_l*w_l, C_l); sizes[l] = (h_l, w_l); -> (n, sum C_l, H, W) channels-last.  None when
    outside the kernel's domain (CPU tensors, more than 4 levels, channel counts not multiples of 8, mixed dtypes)."""
    m0 = token_maps[0]
    if not (m0.is_cuda and 1 <= len(token_maps) <= 4 and m0.dtype in _DT
            and all(m.dtype == m0.dtype and m.shape[2] % 8 == 0 and m.shape[0] == m0.shape[0] for m in token_maps)):
        return None
    maps = [m if m.is_contiguous() else m.contiguous() for m in token_maps]
    return _UpCat.apply(tuple(int(v) for v in out_size), [tuple(int(v) for v in s) for s in sizes], *maps)
