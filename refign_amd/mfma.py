"""Host side of the hand-written matrix-core kernels (csrc/mfma_gemm.hip, csrc/attn.hip).

  gemm_nt / gemm_tn     the three GEMMs of a token-wise Linear (mix_transformer.py:96-103,137-164)
  attention             MiT's efficient self-attention core softmax(scale q k^T) v (mix_transformer.py:150-160),
                        forward + backward, as a torch.autograd.Function over the C ABI

16-bit operands only (bf16 under the step's autocast, f16 inside align()); fp32 callers stay on their own path -- these
functions return None for anything outside the kernels' domain and raise if the HIP library is missing.
"""
import ctypes
import os

import torch

from . import _lib
from ._tensor import current_stream, on_device, ptr

_DT16 = {torch.bfloat16: 1, torch.float16: 2}
ENABLED = True            # (module switch for tests: False = library GEMM / SDPA everywhere)

# Every dense op of a HIP tensor that ends up in a ROCm LIBRARY (hipBLASLt / MIOpen / fused SDPA behind F.linear, torch.mm,
# F.conv2d, scaled_dot_product_attention) instead of a hand-written kernel is recorded here by its call site:
# (kind, dtype, shape signature) -> number of calls.  bench.py prints the table as `config.library_fallbacks`: in its
# default (reduced-precision) mode the only entries allowed are the ones DESIGN.md lists.  fp32 parity mode runs fp32
# operands, for which no matrix-core kernel exists -- those calls are recorded too (dtype float32).
LIBRARY_CALLS = {}
_NOTE = True


def note_library(kind, *tensors):
    """Record a dense library call made for HIP tensors (no-op for CPU tensors: host-side unit tests)."""
    if not _NOTE or not tensors or not tensors[0].is_cuda:
        return
    key = (kind, str(tensors[0].dtype).replace("torch.", ""), tuple(tuple(t.shape) for t in tensors))
    n = LIBRARY_CALLS.get(key)
    if n is None and len(LIBRARY_CALLS) < 4096:
        if os.environ.get("RFN_LOG_LIBRARY", "0") == "1":
            print(f"[refign_amd] library fallback: {kind} {key[1]} {key[2]}", flush=True)
        LIBRARY_CALLS[key] = 1
    elif n is not None:
        LIBRARY_CALLS[key] = n + 1


def library_summary():
    """{kind/dtype: {"calls": n, "shapes": distinct signatures}} of the calls recorded so far."""
    out = {}
    for (kind, dt, _), n in LIBRARY_CALLS.items():
        e = out.setdefault(f"{kind}/{dt}", {"calls": 0, "shapes": 0})
        e["calls"] += n
        e["shapes"] += 1
    return out


def _row_major(t):
    return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.stride(0) >= t.shape[1] \
        and t.data_ptr() % 16 == 0


def gemm_nt(x, w, bias=None, res=None, rowscale=None, rows_per_sample=0, act=0, out=None):
    """y[M,N] = res + rowscale[m // rows_per_sample] * act(x[M,K] @ w[N,K]^T + bias); None if outside the domain."""
    if not (ENABLED and x.is_cuda and x.dtype in _DT16 and w.dtype == x.dtype and _row_major(x) and _row_major(w)):
        return None
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K or K % 64 != 0 or N % 8 != 0 or M == 0:
        return None
    if bias is not None and not (bias.dtype == x.dtype and bias.is_contiguous() and bias.numel() == N):
        return None
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    if res is not None and not (res.dtype == x.dtype and res.shape == out.shape and res.stride() == out.stride()):
        return None
    if rowscale is not None and (rows_per_sample <= 0 or rowscale.dtype != torch.float32 or not rowscale.is_contiguous()):
        return None
    lib = _lib.load_library()
    with on_device(x.device):
        rc = lib.rfn_gemm_nt(ptr(x), ptr(w), ptr(bias), ptr(res), ptr(rowscale), int(rows_per_sample), int(act),
                             ptr(out), M, N, K, x.stride(0), w.stride(0), out.stride(0), _DT16[x.dtype],
                             current_stream(x.device))
    _lib.check(rc, "gemm_nt")
    return out


def pack_conv_weight(w, dtype, scale=None):
    """(N, C, KH, KW) conv weight -> (N, roundup(KH*KW*Cp, 64)) rows of [tap][channel], Cp = C rounded up to 8, zero
    padded: the W operand of rfn_conv2d_nhwc.  `scale` (N,), if given, multiplies the rows (folded BatchNorm)."""
    N, C, KH, KW = w.shape
    Cp = -(-C // 8) * 8
    K = KH * KW * Cp
    Kp = -(-K // 64) * 64
    wf = w.float() if scale is None else w.float() * scale.float().view(-1, 1, 1, 1)
    out = torch.zeros((N, Kp), dtype=dtype, device=w.device)
    t = torch.zeros((N, KH, KW, Cp), dtype=torch.float32, device=w.device)
    t[..., :C] = wf.permute(0, 2, 3, 1)
    out[:, :K] = t.reshape(N, K).to(dtype)
    return out


def conv2d_nhwc(x, wp, bias, KH, KW, stride=1, pad=0, dil=1, act=0, res=None, out=None):
    """x (B, H, W, C) channels-last, C % 8 == 0; wp from pack_conv_weight; -> (B, OH, OW, N) (or into `out`, which may be
    a channel slice of a wider buffer).  None if outside the kernel's domain."""
    if not (ENABLED and x.is_cuda and x.dtype in _DT16 and wp.dtype == x.dtype and x.dim() == 4 and x.is_contiguous()
            and x.shape[-1] % 8 == 0 and wp.is_contiguous()):
        return None
    B, H, W, C = x.shape
    N = wp.shape[0]
    if N % 8 != 0 or wp.shape[1] < KH * KW * C or wp.shape[1] % 64 != 0:
        return None
    OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    if out is None:
        out = torch.empty((B, OH, OW, N), dtype=x.dtype, device=x.device)
    if tuple(out.shape) != (B, OH, OW, N) or out.stride(3) != 1 or out.stride(2) % 8 != 0 or \
            out.stride(1) != OW * out.stride(2) or out.stride(0) != OH * out.stride(1):
        return None
    if res is not None and (res.dtype != x.dtype or res.shape != out.shape or res.stride() != out.stride()):
        return None
    if bias is not None and not (bias.dtype == x.dtype and bias.is_contiguous() and bias.numel() == N):
        return None
    lib = _lib.load_library()
    with on_device(x.device):
        rc = lib.rfn_conv2d_nhwc(ptr(x), ptr(wp), ptr(bias), ptr(res), int(act), ptr(out), B, H, W, C, N, KH, KW, stride,
                                 pad, dil, wp.stride(0), out.stride(2), _DT16[x.dtype], current_stream(x.device))
    _lib.check(rc, "conv2d_nhwc")
    return out


def conv2d_nhwc_dgrad(gy, wt, H, W, C, KH, KW, stride=1, pad=0, dil=1):
    """Data gradient of conv2d_nhwc.  gy (B, OH, OW, N) channels-last contiguous, wt (C, >= roundup(KH*KW*N, 64)) rows of
    [tap][n] (the filter's permute(1, 2, 3, 0), zero padded) -> (B, H, W, C).  None outside the kernel's domain."""
    if not (ENABLED and gy.is_cuda and gy.dtype in _DT16 and wt.dtype == gy.dtype and gy.dim() == 4 and gy.is_contiguous()
            and wt.is_contiguous() and stride & (stride - 1) == 0):
        return None
    B, OH, OW, N = gy.shape
    if N % 8 or C % 8 or wt.shape[0] != C or wt.shape[1] < KH * KW * N or wt.shape[1] % 64:
        return None
    if (OH, OW) != ((H + 2 * pad - dil * (KH - 1) - 1) // stride + 1, (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1):
        return None
    dx = torch.empty((B, H, W, C), dtype=gy.dtype, device=gy.device)
    with on_device(gy.device):
        rc = _lib.load_library().rfn_conv2d_nhwc_dgrad(ptr(gy), ptr(wt), ptr(dx), B, H, W, C, N, KH, KW, stride, pad, dil,
                                                       wt.stride(0), C, _DT16[gy.dtype], current_stream(gy.device))
    _lib.check(rc, "conv2d_nhwc_dgrad")
    return dx


def conv2d_nhwc_wgrad(gy, x, KH, KW, Kpad, stride=1, pad=0, dil=1, bias_out=None):
    """Weight gradient of conv2d_nhwc in the packed layout.  gy (B, OH, OW, N), x (B, H, W, C) channels-last contiguous ->
    fp32 slab partials (S, N, Kpad) whose sum over S is dW[n][(ky, kx, c)] (columns past KH*KW*C are zero); `bias_out` (N,)
    fp32, if given, += column sums of gy.  None outside the kernel's domain."""
    if not (ENABLED and gy.is_cuda and gy.dtype in _DT16 and x.dtype == gy.dtype and gy.dim() == 4 and x.dim() == 4
            and gy.is_contiguous() and x.is_contiguous()):
        return None
    B, OH, OW, N = gy.shape
    _, H, W, C = x.shape
    if N % 64 or C % 2 or Kpad % 64 or Kpad < KH * KW * C or Kpad >= 65536 or x.shape[0] != B:
        return None
    if (OH, OW) != ((H + 2 * pad - dil * (KH - 1) - 1) // stride + 1, (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1):
        return None
    T = B * OH * OW
    tile = 128 if (N % 128 == 0 and Kpad % 128 == 0) else 64
    rows = slab_rows(T, (N // tile) * (Kpad // tile))
    S = -(-T // rows)
    part = torch.empty((S, N, Kpad), dtype=torch.float32, device=gy.device)
    with on_device(gy.device):
        rc = _lib.load_library().rfn_conv2d_nhwc_wgrad(ptr(gy), ptr(x), ptr(part), ptr(bias_out), B, H, W, C, N, KH, KW, stride,
                                                       pad, dil, N, Kpad, rows, 0, _DT16[gy.dtype], current_stream(gy.device))
    _lib.check(rc, "conv2d_nhwc_wgrad")
    return part


_TN_WGS = 1024      # target workgroups of a weight-gradient launch (swept in round 3: 1024 / 512 / 384 / 256 -> 185.4 / 185.2 /
#                     186.4 / 189.6 ms per step)


_TN_ATOMICS = 2500000     # atomic adds per launch (see slab_rows)
_TN_MIN_WGS = 512


def slab_rows(T, tiles, nk=0):
    """Rows per slab of the split-T weight-gradient GEMM: enough slabs to fill the chip (~1024 workgroups with the
    output tiles), slabs of at least 256 rows, multiples of 32 -- and not more slabs than ~2.5 M fp32 atomics in all: every
    slab ADDS a whole N x K partial, and the L2 retires ~0.6 T atomic adds per second (tools: a 320 x 320 gradient of 8160
    rows takes 16.2 us in 32 slabs, 12.1 us in 10; 27 % of a 320 x 1280 launch is its atomics)."""
    want = max(1, min(64, _TN_WGS // max(tiles, 1)))
    if nk > 0:
        want = min(want, max(4, _TN_ATOMICS // nk, -(-_TN_MIN_WGS // max(tiles, 1))))     # ... but at least ~512 workgroups
    rows = -(-T // want)
    rows = max(256, -(-rows // 32) * 32)
    return rows


def gemm_tn(g, x, rows_per_slab=None, out=None, bias_out=None, rowscale=None, rows_per_sample=0):
    """(diag(rowscale) g[T,N])^T @ x[T,K] with the token dimension split into slabs (`rowscale`: fp32, one value per
    `rows_per_sample` consecutive rows -- the stochastic-depth scale of the branch the gradient g belongs to).  `out` None: returns the fp32 partials (S, N, K)
    (deterministic; sum over S is the result).  `out` (N, K) fp32: every slab is ADDED into it with fp32 atomics (the
    parameter's view of the flat gradient buffer) and `bias_out` (N,) fp32, if given, += the column sums of g.
    None if outside the kernel's domain."""
    if not (ENABLED and g.is_cuda and g.dtype in _DT16 and x.dtype == g.dtype and g.dim() == 2 and x.dim() == 2
            and g.stride(1) == 1 and x.stride(1) == 1 and g.stride(0) % 2 == 0 and x.stride(0) % 2 == 0
            and g.data_ptr() % 4 == 0 and x.data_ptr() % 4 == 0):
        return None
    T, N = g.shape
    K = x.shape[1]
    if x.shape[0] != T or N % 64 != 0 or K % 64 != 0 or T == 0:
        return None
    if rowscale is not None and not (rowscale.dtype == torch.float32 and rowscale.is_contiguous() and rows_per_sample > 0
                                     and rowscale.numel() * rows_per_sample >= T):
        return None
    if rows_per_slab is None:
        tile = 128 if (N % 128 == 0 and K % 128 == 0) else 64
        rows_per_slab = slab_rows(T, (N // tile) * (K // tile), N * K)
    S = -(-T // rows_per_slab)
    if out is not None:
        if not (out.dtype == torch.float32 and out.is_contiguous() and out.numel() == N * K and
                (bias_out is None or (bias_out.dtype == torch.float32 and bias_out.is_contiguous()
                                      and bias_out.numel() == N))):
            return None
        part = out
    else:
        part = torch.empty((S, N, K), dtype=torch.float32, device=g.device)
    lib = _lib.load_library()
    with on_device(g.device):
        rc = lib.rfn_gemm_tn(ptr(g), ptr(x), ptr(part), T, N, K, g.stride(0), x.stride(0), int(rows_per_slab),
                             0 if out is None else 1, ptr(bias_out) if out is not None else None, ptr(rowscale),
                             int(rows_per_sample), _DT16[g.dtype], current_stream(g.device))
    _lib.check(rc, "gemm_tn")
    return part


# ---------------------------------------------------------------------------------------------------------------------
# deferred weight gradients (round 5): a backward pass's weight gradients are off its dependency chain -- only the data
# gradients feed the next layer -- but launched where autograd reaches them they sit IN the chain: six latency-bound launches
# of 15-25 us per MiT block between the data-gradient GEMMs.  Inside `deferred_wgrads()` (the trainer's backward) the
# accumulate-form calls of Linear layers are queued with their operands kept alive and handed to the grouped kernel
# (rfn_gemm_tn_grouped, up to 8 problems per launch, 64 x 64 tiles) at the marks the MiT blocks leave in the autograd graph and
# at the end of the pass.  Same products, same fp32 atomics into the flat gradient buffer; only the launch structure changes.
# ---------------------------------------------------------------------------------------------------------------------
GROUP_WGRADS = True           # (module switch: tests compare the grouped step with the one-launch-per-gradient step)
_TN_GROUP_MAX = 8
_WGRAD_QUEUE = None          # None: not deferring; else a list of (g, x, out, bias_out, rowscale, rows_per_sample, rows_per_slab)


def _tn_domain(g, x, rowscale, rows_per_sample):
    if not (ENABLED and g.is_cuda and g.dtype in _DT16 and x.dtype == g.dtype and g.dim() == 2 and x.dim() == 2
            and g.stride(1) == 1 and x.stride(1) == 1 and g.stride(0) % 2 == 0 and x.stride(0) % 2 == 0
            and g.data_ptr() % 4 == 0 and x.data_ptr() % 4 == 0):
        return False
    T, N = g.shape
    K = x.shape[1]
    if x.shape[0] != T or N % 64 != 0 or K % 64 != 0 or T == 0:
        return False
    if rowscale is not None and not (rowscale.dtype == torch.float32 and rowscale.is_contiguous() and rows_per_sample > 0
                                     and rowscale.numel() * rows_per_sample >= T):
        return False
    return True


def defer_gemm_tn(g, x, out, bias_out=None, rowscale=None, rows_per_sample=0):
    """Queue `out += (diag(rowscale) g)^T x` (+ `bias_out += column sums of g`) for the next flush.  False when nothing is being
    deferred or the problem is outside the grouped kernel's domain (the caller then launches it where it stands)."""
    q = _WGRAD_QUEUE
    if q is None or not _tn_domain(g, x, rowscale, rows_per_sample):
        return False
    T, N = g.shape
    K = x.shape[1]
    if not (out.dtype == torch.float32 and out.is_contiguous() and out.numel() == N * K and
            (bias_out is None or (bias_out.dtype == torch.float32 and bias_out.is_contiguous() and bias_out.numel() == N))):
        return False
    if g.stride(0) % 8 or x.stride(0) % 8 or g.data_ptr() % 16 or x.data_ptr() % 16:
        return False
    rows = slab_rows(T, (N // 64) * (K // 64), N * K)
    if rowscale is not None and (rows + rows_per_sample - 1) // rows_per_sample + 1 > 64:
        return False
    q.append((g, x, out, bias_out, rowscale, int(rows_per_sample), int(rows)))
    return True


def flush_wgrads():
    """Launch what has been queued: one grouped launch per (with / without row scale, dtype) class and 8 problems."""
    q = _WGRAD_QUEUE
    if not q:
        return
    items, q[:] = list(q), []
    classes = {}
    for it in items:
        classes.setdefault((it[4] is not None, it[0].dtype, it[0].device), []).append(it)
    lib = _lib.load_library()
    vp, lg, it_ = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
    for (seg, dt, dev), lst in classes.items():
        for a in range(0, len(lst), _TN_GROUP_MAX):
            grp = lst[a:a + _TN_GROUP_MAX]
            n = len(grp)
            arr = lambda ty, vals: (ty * n)(*vals)  # noqa: E731
            G = arr(vp, [t[0].data_ptr() for t in grp])
            X = arr(vp, [t[1].data_ptr() for t in grp])
            P = arr(vp, [t[2].data_ptr() for t in grp])
            Bv = arr(vp, [None if t[3] is None else t[3].data_ptr() for t in grp])
            R = arr(vp, [None if t[4] is None else t[4].data_ptr() for t in grp])
            T = arr(lg, [t[0].shape[0] for t in grp])
            N = arr(lg, [t[0].shape[1] for t in grp])
            K = arr(lg, [t[1].shape[1] for t in grp])
            ldg = arr(lg, [t[0].stride(0) for t in grp])
            ldx = arr(lg, [t[1].stride(0) for t in grp])
            rps = arr(it_, [t[6] for t in grp])
            rsa = arr(it_, [t[5] for t in grp])
            cast = lambda a_: ctypes.cast(a_, vp)  # noqa: E731
            with on_device(dev):
                rc = lib.rfn_gemm_tn_grouped(n, cast(G), cast(X), cast(P), cast(Bv), cast(R), cast(T), cast(N), cast(K),
                                             cast(ldg), cast(ldx), cast(rps), cast(rsa), _DT16[dt], current_stream(dev))
            _lib.check(rc, "gemm_tn_grouped")


class deferred_wgrads:
    """Context manager around a backward pass: weight gradients of Linear layers are queued and launched in groups (at the
    block marks, `wgrad_mark`, and on exit).  Re-entrant use keeps the outer queue."""

    def __enter__(self):
        global _WGRAD_QUEUE
        self._outer = _WGRAD_QUEUE
        if GROUP_WGRADS and ENABLED and self._outer is None:
            _WGRAD_QUEUE = []
        return self

    def __exit__(self, *exc):
        global _WGRAD_QUEUE
        if self._outer is None and _WGRAD_QUEUE is not None:
            try:
                if exc[0] is None:
                    flush_wgrads()
            finally:
                _WGRAD_QUEUE = None
        return False


class _WgradMark(torch.autograd.Function):
    """Identity whose backward flushes the queued weight gradients: placed on a block's INPUT, it runs when the backward pass
    has left the block."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        # (flushing at every 4th mark or once per pass measured 0.2-0.4 ms better than per block on one box and within noise on
        # another, profiles/r05_vgpr_form_ab.txt: per block keeps the operands' lifetime short)
        flush_wgrads()
        return g


def wgrad_mark(x):
    return _WgradMark.apply(x) if (GROUP_WGRADS and torch.is_grad_enabled() and x.requires_grad) else x


# ---------------------------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------------------------
_PACK_BLOCK = 4096


def _pack(src, batch_stride, row_stride, B, heads, nrows, nblk, want_r=True, want_t=True, src2=None):
    """R- / T-packs of `src` (and of `src2`, same geometry, in the same launch): ((r, t), (r2, t2))"""
    dev = src.device
    nbytes = B * heads * nblk * _PACK_BLOCK
    new = lambda want: torch.empty(nbytes, dtype=torch.uint8, device=dev) if want else None  # noqa: E731
    r, t = new(want_r), new(want_t)
    r2, t2 = (new(want_r), new(want_t)) if src2 is not None else (None, None)
    lib = _lib.load_library()
    with on_device(dev):
        rc = lib.rfn_attn_pack(ptr(src), batch_stride, row_stride, B, heads, nrows, nblk, ptr(r), ptr(t), ptr(src2),
                               ptr(r2), ptr(t2), current_stream(dev))
    _lib.check(rc, "attn_pack")
    return (r, t), (r2, t2)


def _attn_ok(q, kv, heads):
    if not (ENABLED and q.is_cuda and q.dtype in _DT16 and kv.dtype == q.dtype and q.dim() == 3 and kv.dim() == 3):
        return False
    B, N, C = q.shape
    return (C == heads * 64 and kv.shape[0] == B and kv.shape[2] == 2 * C and q.is_contiguous() and kv.is_contiguous()
            and q.data_ptr() % 16 == 0 and kv.data_ptr() % 16 == 0)


def _dims(q, kv):
    B, N, C = q.shape
    Nkv = kv.shape[1]
    nkblk = -(-Nkv // 32)
    nkblk += nkblk & 1                                  # forward / dQ walk the keys two blocks per stage
    nqblk = -(-N // 32)
    return B, N, C, Nkv, nkblk, nqblk, nqblk * 32


def _fwd(q, kv, heads, scale, need_bwd):
    B, N, C, Nkv, nkblk, nqblk, nqpad = _dims(q, kv)
    dev = q.device
    # K and V in one launch: the kv tensor is (B, Nkv, 2 * heads, 64), i.e. 2 * heads "heads"; V's head h is pack head
    # heads + h.  The forward needs K's R-pack and V's T-pack, the backward also V's R-pack and K's T-pack.
    (kvr, kvt), _ = _pack(kv, kv.stride(0), kv.stride(1), B, 2 * heads, Nkv, nkblk)
    voff = heads * nkblk * _PACK_BLOCK
    o = torch.empty_like(q)
    lse2 = torch.empty((B * heads, nqpad), dtype=torch.float32, device=dev)
    lib = _lib.load_library()
    with on_device(dev):
        rc = lib.rfn_attn_fwd(ptr(q), q.stride(0), q.stride(1), kvr.data_ptr(), kvt.data_ptr() + voff, ptr(o),
                              o.stride(0), o.stride(1), ptr(lse2), B, heads, N, Nkv, nkblk, nqpad, float(scale),
                              2 * heads, _DT16[q.dtype], current_stream(dev))
    _lib.check(rc, "attn_fwd")
    return o, lse2, (kvr, kvt)


def _chunk_blocks(nqblk, Nkv, BH):
    """32-query blocks per dK/dV workgroup: at most 256 workgroups (512 threads each, ONE per CU at a time), at least 4
    blocks each.  A workgroup's time is mostly fixed cost (its keys' K / V rows, the first stage, 64 atomics per lane):
    320 workgroups are two rounds over the 256 CUs and cost a whole second round -- measured on the student's shapes
    (tools/attn_bench.py, backward of B=4: stage 1 / 2 / 3): 157 / 125 / 98 us at a target of 512, 156 / 116 / 95 at 384,
    132 / 100 / 80 at 256, 175 / 144 / 125 at 768.  More, shorter chunks also add fp32 atomics on the dK / dV image (every
    chunk adds its 256 x 64 x 2 partial).  Those numbers are from isolated launches; INSIDE the step, where three streams share the
    device and the L2 atomic units, fewer and longer chunks win: step 154.1 / 153.6 / 153.0 ms at targets of 256 / 192 / 128,
    155.8 at 384; 151.4 / 151.4 / 152.6 / 153.1 at 128 / 96 / 64 / 48 (another box) -- 128 is the default since."""
    kblocks = -(-Nkv // 256)
    chunks = max(1, 128 // max(1, kblocks * BH))
    return max(4, -(-nqblk // chunks))


class _AttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kv, heads, scale):
        need = q.requires_grad or kv.requires_grad
        o, lse2, packs = _fwd(q, kv, heads, scale, need)
        if need:
            ctx.save_for_backward(q, kv, o, lse2, *packs)
            ctx.heads, ctx.scale = heads, scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, o, lse2, kvr, kvt = ctx.saved_tensors
        heads, scale = ctx.heads, ctx.scale
        B, N, C, Nkv, nkblk, nqblk, nqpad = _dims(q, kv)
        dev = q.device
        if do.dtype != q.dtype:
            do = do.to(q.dtype)
        if not do.is_contiguous():
            do = do.contiguous()
        dt = _DT16[q.dtype]
        lib = _lib.load_library()
        dq = torch.empty_like(q)
        delta = torch.empty_like(lse2)
        voff = heads * nkblk * _PACK_BLOCK
        with on_device(dev):
            rc = lib.rfn_attn_bwd_dq(ptr(q), q.stride(0), q.stride(1), ptr(do), ptr(o), o.stride(0), o.stride(1),
                                     kvr.data_ptr(), kvr.data_ptr() + voff, kvt.data_ptr(), ptr(lse2), ptr(delta),
                                     ptr(dq), dq.stride(0), dq.stride(1), B, heads, N, Nkv, nkblk, nqpad, float(scale),
                                     2 * heads, dt, current_stream(dev))
        _lib.check(rc, "attn_bwd_dq")
        (qr, qt), (gr, gt) = _pack(q, q.stride(0), q.stride(1), B, heads, N, nqblk, src2=do)
        nkpad = -(-Nkv // 32) * 32
        accT = torch.empty(B * heads * 2 * 64 * nkpad, dtype=torch.float32, device=dev)
        dkv = torch.empty_like(kv)
        k_view, v_view = kv[:, :, :C], kv[:, :, C:]
        with on_device(dev):
            rc = lib.rfn_attn_bwd_dkv(ptr(k_view), ptr(v_view), kv.stride(0), kv.stride(1), ptr(qr), ptr(qt), ptr(gr),
                                      ptr(gt), ptr(lse2), ptr(delta), ptr(accT), ptr(dkv), B, heads, N, Nkv, nqblk,
                                      nqpad, nkpad, _chunk_blocks(nqblk, Nkv, B * heads), float(scale), dt,
                                      current_stream(dev))
        _lib.check(rc, "attn_bwd_dkv")
        return dq, dkv, None, None


def attention(q, kv, heads, scale):
    """q: (B, N, heads*64) output of the q Linear; kv: (B, Nkv, 2*heads*64) output of the kv Linear (K then V, each
    (heads, 64) per token -- the layout `.reshape(B, -1, 2, heads, 64)` of mix_transformer.py:147-149 reads).
    Returns (B, N, heads*64) = `(attn @ v).transpose(1, 2).reshape(B, N, C)`, or None if outside the kernels' domain."""
    if not _attn_ok(q, kv, heads):
        return None
    return _AttnFn.apply(q, kv, heads, scale)
