"""K5 -- fp8 (OCP e4m3) matrix-core path of the gradient-free EMA teacher (csrc/f8.hip).

BASELINE.json config 5 ("bf16 HRDA + fp8 MFMA attention").  No reference analogue: the reference trains with 16-bit AMP
(README.md:262); what runs here is its EMA-teacher forward (segmentation_model.py:204-209: MiT-B5 on the HRDA views of
(target, reference), mix_transformer.py:79-207) with every token-wise Linear and the attention core on the gfx950 fp8
matrix instruction.  Student, decode heads, matcher, losses: unchanged (bf16 / fp16 / fp32 as in the default mode).

One MiT block in this mode (`block_forward`), residual stream bf16:
    LayerNorm -> e4m3 | q GEMM -> e4m3 | [patchify -> sr GEMM -> bf16 -> LayerNorm -> e4m3] | kv GEMM -> e4m3 -> packs
    attention (fp32 softmax, e4m3 P) -> e4m3 | proj GEMM + residual (+ stochastic-depth scale) -> bf16
    LayerNorm -> e4m3 | fc1 GEMM -> e4m3 | depthwise 3x3 + GELU -> e4m3 | fc2 GEMM + residual -> bf16
Activations are quantised by their PRODUCER (LayerNorm / GEMM / attention / depthwise epilogues) with ONE power-of-two
scale ACT_Q; weights carry one fp32 scale per output row and are re-quantised once per step after the EMA update
(`requantize`, one multi-tensor launch).  The product path has no fallback: outside the kernels' domain it raises.
"""
import contextlib
import os

import numpy as np
import torch

from . import _lib, params
from ._tensor import current_stream, on_device, ptr

ACT_Q = 8.0                 # stored byte = e4m3(value * ACT_Q): |value| < 56 in range, subnormal below 2e-3
_ACTIVE = [False]


def active():
    return _ACTIVE[0]


@contextlib.contextmanager
def teacher_f8(on=True):
    """Inside: MiT blocks of gradient-free passes take the fp8 path (seg.Block.forward)."""
    old = _ACTIVE[0]
    _ACTIVE[0] = bool(on)
    try:
        yield
    finally:
        _ACTIVE[0] = old


def _u8(shape, dev):
    return torch.empty(shape, dtype=torch.uint8, device=dev)


def quantize(x, q=ACT_Q):
    """bf16 tensor -> e4m3 bytes of x * q (saturating, round to nearest even)."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.numel() % 4 == 0
    y = _u8(x.shape, x.device)
    with on_device(x.device):
        rc = _lib.load_library().rfn_quant_f8(ptr(x), ptr(y), x.numel(), float(q), current_stream(x.device))
    _lib.check(rc, "quant_f8")
    return y


# ---------------------------------------------------------------------------------------------------------------------
# weights: e4m3 rows + fp32 row scales in ONE uint8 buffer per parameter, re-filled in place (captured graphs keep pointing
# at it) from the cached bf16 copy of the parameter (params.derived), which params.refresh() re-fills in place first
# ---------------------------------------------------------------------------------------------------------------------
_WEIGHTS = {}        # id(param) -> dict(param=, src=, buf=, N=, K=)
_TABLE = {}          # device -> (signature, table tensor, nchunks)


def _src_2d(p, kind):
    if kind == "linear":
        return params.as_dtype(p, torch.bfloat16).view(p.shape[0], -1)
    if kind == "patch":                                           # sr conv as a Linear over (ry, rx, c) patches
        Co = p.shape[0]
        return params.derived(p, (torch.bfloat16, "patch_linear"), lambda t: t.to(torch.bfloat16).permute(0, 2, 3, 1).contiguous(),
                              lambda t: t.permute(0, 2, 3, 1)).view(Co, -1)
    raise ValueError(kind)


def weight(p, kind="linear"):
    """(w8 (N, K) uint8 view, scales (N,) fp32 view) of parameter `p`, quantised on first use."""
    ent = _WEIGHTS.get(id(p))
    src = _src_2d(p, kind)
    if ent is None or ent["param"] is not p or ent["src"].data_ptr() != src.data_ptr():
        N, K = src.shape
        if K % 16 or N % 16:
            raise RuntimeError(f"f8.weight: shape {tuple(src.shape)} outside the fp8 kernels' domain (N, K % 16)")
        buf = torch.empty(N * K + 4 * N, dtype=torch.uint8, device=src.device)
        ent = _WEIGHTS[id(p)] = dict(param=p, src=src, buf=buf, N=N, K=K, kind=kind)
        _TABLE.pop(src.device, None)
        _quantize_entries([ent])
    N, K = ent["N"], ent["K"]
    return ent["buf"][:N * K].view(N, K), ent["buf"][N * K:].view(torch.float32)


def _rows(ent):
    N, K = ent["N"], ent["K"]
    sp, dp = ent["src"].data_ptr(), ent["buf"].data_ptr()
    return [(sp + 2 * K * r, dp + K * r, dp + N * K + 4 * r, K | (min(4, N - r) << 32)) for r in range(0, N, 4)]


def _launch(table, n, dev):
    with on_device(dev):
        rc = _lib.load_library().rfn_quant_rows_f8(ptr(table), n, current_stream(dev))
    _lib.check(rc, "quant_rows_f8")


def _quantize_entries(ents):
    dev = ents[0]["src"].device
    rows = [r for e in ents for r in _rows(e)]
    table = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)
    _launch(table, len(rows), dev)
    return table, len(rows)


def requantize():
    """Re-quantise every registered weight from its (already refreshed) bf16 copy: one launch per device.  Call after the
    EMA update of a step (uda.update_momentum_encoder)."""
    by_dev = {}
    for key, e in list(_WEIGHTS.items()):
        src = _src_2d(e["param"], e["kind"])
        if src.data_ptr() != e["src"].data_ptr():           # the bf16 copy moved (parameter re-allocated): re-register
            del _WEIGHTS[key]
            _TABLE.pop(e["src"].device, None)
            continue
        by_dev.setdefault(src.device, []).append(e)
    for dev, ents in by_dev.items():
        sig = (len(ents), ents[0]["buf"].data_ptr(), ents[-1]["buf"].data_ptr())
        t = _TABLE.get(dev)
        if t is None or t[0] != sig:
            table, n = _quantize_entries(ents)
            _TABLE[dev] = (sig, table, n)
        else:
            _launch(t[1], t[2], dev)


def reset():
    _WEIGHTS.clear()
    _TABLE.clear()


# ---------------------------------------------------------------------------------------------------------------------
# kernels
# ---------------------------------------------------------------------------------------------------------------------
def gemm_nt(x8, w8, ws, bias=None, res=None, rowscale=None, rows_per_sample=0, act=0, out_f8=False, x_scale=1.0 / ACT_Q,
            out_q=ACT_Q):
    """x8 (M, K) e4m3 bytes, w8 (N, K), ws (N,) fp32 -> bf16 (M, N) [res + rowscale * (...)] or e4m3 (M, N) of (...) * out_q."""
    M, K = x8.shape
    N = w8.shape[0]
    if not (x8.dtype == torch.uint8 and w8.dtype == torch.uint8 and x8.stride(1) == 1 and w8.stride(1) == 1
            and w8.shape[1] == K and K % 16 == 0 and N % 16 == 0 and x8.stride(0) % 16 == 0 and w8.stride(0) % 16 == 0
            and x8.data_ptr() % 16 == 0 and w8.data_ptr() % 16 == 0):
        raise RuntimeError(f"f8.gemm_nt: operands outside the kernel's domain: x {tuple(x8.shape)} w {tuple(w8.shape)}")
    dev = x8.device
    y = torch.empty((M, N), dtype=torch.uint8 if out_f8 else torch.bfloat16, device=dev)
    if res is not None:
        assert not out_f8 and res.dtype == torch.bfloat16 and res.shape == y.shape and res.is_contiguous()
    if bias is not None:
        assert bias.dtype == torch.bfloat16 and bias.is_contiguous() and bias.numel() == N
    with on_device(dev):
        rc = _lib.load_library().rfn_gemm_nt_f8(ptr(x8), ptr(w8), ptr(ws), float(x_scale), ptr(bias), ptr(res), ptr(rowscale),
                                                int(rows_per_sample), int(act), ptr(y), 1 if out_f8 else 0, float(out_q),
                                                M, N, K, x8.stride(0), w8.stride(0), y.stride(0), current_stream(dev))
    _lib.check(rc, "gemm_nt_f8")
    return y


def layernorm(x, ln, out_q=ACT_Q):
    """LayerNorm module `ln` on bf16 rows -> e4m3 bytes, same shape."""
    C = x.shape[-1]
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and C % 8 == 0
    y = _u8(x.shape, x.device)
    w32 = params.as_dtype(ln.weight, torch.float32).detach()
    b32 = params.as_dtype(ln.bias, torch.float32).detach()
    with on_device(x.device):
        rc = _lib.load_library().rfn_layernorm_fwd_f8(ptr(x), ptr(w32), ptr(b32), ptr(y), x.numel() // C, C, float(ln.eps),
                                                      float(out_q), current_stream(x.device))
    _lib.check(rc, "layernorm_fwd_f8")
    return y


def dwconv_gelu(h8, dw, B, H, W, x_scale=1.0 / ACT_Q, out_q=ACT_Q):
    """gelu(depthwise3x3(h) + bias) on e4m3 tokens (B, H*W, C) -> e4m3, `dw` = the nn.Conv2d(C, C, 3, groups=C)."""
    C = h8.shape[-1]
    w_tap = params.derived(dw.weight, "tap_major_f32", lambda t: t.float().reshape(C, 9).t().contiguous(),
                           lambda t: t.reshape(C, 9).t())
    b32 = None if dw.bias is None else params.as_dtype(dw.bias, torch.float32).detach()
    y = _u8(h8.shape, h8.device)
    with on_device(h8.device):
        rc = _lib.load_library().rfn_dwconv3x3_gelu_nhwc_fwd_f8(ptr(h8), ptr(w_tap), ptr(b32), ptr(y), B, H, W, C,
                                                                float(x_scale), float(out_q), current_stream(h8.device))
    _lib.check(rc, "dwconv3x3_gelu_nhwc_fwd_f8")
    return y


def attention(q8, kv8, heads, scale, q_scale=1.0 / ACT_Q, kv_scale=1.0 / ACT_Q, out_q=ACT_Q):
    """q8 (B, N, heads*64), kv8 (B, Nkv, 2*heads*64) e4m3 -> o8 (B, N, heads*64) e4m3 of softmax(scale q k^T) v * out_q."""
    B, N, C = q8.shape
    Nkv = kv8.shape[1]
    assert C == heads * 64 and kv8.shape[2] == 2 * C and q8.is_contiguous() and kv8.is_contiguous()
    dev = q8.device
    nst = -(-Nkv // 64)
    pack = _u8(B * heads * nst * 8192, dev)
    o8 = _u8(q8.shape, dev)
    lib = _lib.load_library()
    with on_device(dev):
        rc = lib.rfn_attn_pack_f8(ptr(kv8), kv8.stride(0), kv8.stride(1), B, heads, Nkv, nst, ptr(pack), current_stream(dev))
        _lib.check(rc, "attn_pack_f8")
        rc = lib.rfn_attn_fwd_f8(ptr(q8), q8.stride(0), q8.stride(1), ptr(pack), ptr(o8), o8.stride(0), o8.stride(1), B,
                                 heads, N, Nkv, nst, float(scale), float(q_scale), float(kv_scale), float(kv_scale),
                                 float(out_q), current_stream(dev))
    _lib.check(rc, "attn_fwd_f8")
    return o8


def _patchify8(x8, B, H, W, C, r):
    """(B, H*W, C) e4m3 tokens -> (B*(H/r)*(W/r), r*r*C) patches: the byte mover of the 16-bit path on C/2 "elements"."""
    Hr, Wr = H // r, W // r
    out = _u8((B * Hr * Wr, r * r * C), x8.device)
    with on_device(x8.device):
        rc = _lib.load_library().rfn_patchify_tokens(ptr(x8), ptr(out), B, H, W, C // 2, r, 1, 0, current_stream(x8.device))
    _lib.check(rc, "patchify_tokens")
    return out, Hr, Wr


def block_supported(blk, x):
    C = x.shape[-1]
    a = blk.attn
    return (x.is_cuda and x.dtype == torch.bfloat16 and C % 16 == 0 and C // a.num_heads == 64
            and blk.mlp.fc1.out_features % 16 == 0 and a.attn_drop.p == 0. and a.proj_drop.p == 0. and blk.mlp.drop.p == 0.)


def block_forward(blk, x, H, W, masks32=None):
    """One MiT block (mix_transformer.py:167-207) of a gradient-free pass in fp8; x (B, N, C) bf16 -> (B, N, C) bf16."""
    B, N, C = x.shape
    a, m = blk.attn, blk.mlp
    x = x.contiguous()
    bf = torch.bfloat16
    rs = (lambda i: None) if masks32 is None else (lambda i: masks32[i])
    rps = N
    xn8 = layernorm(x, blk.norm1).view(B * N, C)
    q8 = gemm_nt(xn8, *weight(a.q.weight), bias=params.as_dtype(a.q.bias, bf), out_f8=True)
    if a.sr_ratio > 1:
        r = a.sr_ratio
        p8, Hr, Wr = _patchify8(xn8, B, H, W, C, r)
        red = gemm_nt(p8, *weight(a.sr.weight, "patch"), bias=params.as_dtype(a.sr.bias, bf))
        kvin8 = layernorm(red, a.norm)
        Nkv = Hr * Wr
    else:
        kvin8, Nkv = xn8, N
    kv8 = gemm_nt(kvin8, *weight(a.kv.weight), bias=params.as_dtype(a.kv.bias, bf), out_f8=True)
    o8 = attention(q8.view(B, N, C), kv8.view(B, Nkv, 2 * C), a.num_heads, a.scale)
    x = gemm_nt(o8.view(B * N, C), *weight(a.proj.weight), bias=params.as_dtype(a.proj.bias, bf), res=x.view(B * N, C),
                rowscale=rs(0), rows_per_sample=rps if masks32 is not None else 0).view(B, N, C)
    if HYBRID_FFN:
        # the Mix-FFN half on the bf16 kernels: the fused fc1 + depthwise + GELU kernel keeps the 4C-wide hidden tensor on the CU
        # (csrc/mixffn.hip), which is worth more than fp8 operands for fc1 / fc2 were (round 6: the all-fp8 block tied with bf16)
        with teacher_f8(False):
            return m(blk.norm2(x), H, W, res=x, rowscale=rs(1))
    xn8 = layernorm(x, blk.norm2).view(B * N, C)
    h8 = gemm_nt(xn8, *weight(m.fc1.weight), bias=params.as_dtype(m.fc1.bias, bf), out_f8=True)
    g8 = dwconv_gelu(h8.view(B, N, -1), m.dwconv.dwconv, B, H, W)
    return gemm_nt(g8.view(B * N, -1), *weight(m.fc2.weight), bias=params.as_dtype(m.fc2.bias, bf), res=x.view(B * N, C),
                   rowscale=rs(1), rows_per_sample=rps if masks32 is not None else 0).view(B, N, C)


HYBRID_FFN = True             # K5: attention half of a block in fp8, Mix-FFN half on the bf16 kernels (fused front half)
ENV_DEFAULT = False           # K5 is chosen per model (`model.teacher_f8`, bench.py --precision k5), not by the environment
