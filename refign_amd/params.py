"""Parameter plumbing of the training step, laid out for the flat gradient buffer (trainer.FlatGradBuffer).

Two things autograd + autocast do per parameter per pass, which on a MiT-B5 (1 000 parameters, 4 forward and 3 backward
passes per Refign step) add up to ~7 000 launches of tiny kernels (profiles/r01_step_shapes_elem.txt):
  * autocast re-casts every fp32 weight/bias to bf16 at every use              -> `derived()`: one versioned bf16 copy
    per parameter, re-made only after the parameter changed in place (optimizer step, EMA update, load_state_dict);
  * every weight gradient is materialised, cast to fp32 and added to `.grad`   -> `grad_sink()`: the hand-written
    backward kernels (sum_rows, layernorm_bwd, dwconv3x3_bwd_weight) ACCUMULATE straight into the parameter's view of
    the flat fp32 gradient buffer and return None to autograd.
"""
import os

import torch
from torch.optim.optimizer import register_optimizer_step_post_hook

from . import _lib
from ._tensor import current_stream, on_device, ptr, workspace

_DT = {torch.float32: 0, torch.bfloat16: 1}


def derived(p, key, fn, refill=None):
    """fn(p.detach()) cached on the parameter, recomputed when p was modified in place or re-allocated.
    `refill(p.detach())`, if given, is a VIEW of the parameter with the cached tensor's shape: refresh() then updates
    the cached tensor in place with one multi-tensor copy (dtype / layout conversions) instead of dropping it."""
    cache = p.__dict__.setdefault("_rfn_derived", {})
    ent = cache.get(key)
    if ent is None or ent[1] != p._version or ent[2] != p.data_ptr():
        with torch.no_grad():
            t = fn(p.detach())
        cache[key] = ent = (t, p._version, p.data_ptr(), refill)
        _GENERATION[0] += 1                            # cached refresh plans are stale
    return ent[0]


_GENERATION = [0]        # bumped whenever a derived entry is created, replaced or dropped
_PLANS = {}              # plan key -> what refresh() does for one fixed parameter set (see refresh)


def _identity(t):
    return t


def _build_plan(params):
    """One pass over the parameters' caches: which cached tensors are re-filled from which views of their parameter.
    The views alias the parameters' storage, so the same (dst, src) lists serve every later refresh of this parameter
    set for as long as no cache entry is created or dropped and no parameter is re-allocated."""
    entries, dst, src = [], [], []
    for p in params:
        cache = p.__dict__.get("_rfn_derived")
        if not cache:
            continue
        for key in list(cache):
            t, _, dptr, refill = cache[key]
            if refill is not None and dptr == p.data_ptr():
                v = refill(p.detach())
                if v.shape == t.shape:
                    dst.append(t)
                    src.append(v)
                    cache[key] = (t, p._version, dptr, refill)
                    entries.append((cache, key, p))
                    continue
            del cache[key]
            _GENERATION[0] += 1
    # plain fp32 -> bf16 casts of contiguous tensors: ONE launch of the multi-tensor cast kernel (csrc/reduce.hip);
    # torch._foreach_copy_ with a dtype change is one tiny kernel per tensor.  Layout-changing copies stay per tensor.
    fast = [i for i, (d, s_) in enumerate(zip(dst, src))
            if d.is_cuda and d.dtype == torch.bfloat16 and s_.dtype == torch.float32 and d.is_contiguous()
            and s_.is_contiguous() and d.numel() == s_.numel() and d.numel() > 0]
    casts = {}
    if len(fast) >= 4:
        for i in fast:
            casts.setdefault(dst[i].device, []).append(i)
        skip = set(fast)
    else:
        skip = set()
    # transposed bf16 copies (W^T of the Linear weights): one launch of the tile-transpose kernel instead of one strided
    # copy per tensor
    rest = [i for i in range(len(dst)) if i not in skip]
    tr = [i for i in rest if dst[i].is_cuda and dst[i].dtype == torch.bfloat16 and src[i].dtype == torch.float32
          and dst[i].dim() == 2 and dst[i].is_contiguous() and src[i].dim() == 2 and src[i].t().is_contiguous()
          and src[i].numel() > 0]
    transposes = []
    if len(tr) >= 4:
        by_dev = {}
        for i in tr:
            by_dev.setdefault(dst[i].device, []).append(i)
        for dev, idx in by_dev.items():
            transposes.append(_transpose_table([dst[i] for i in idx], [src[i] for i in idx], dev) +
                              (dev, [dst[i] for i in idx], [src[i] for i in idx]))
        skip = skip | set(tr)
    # every other layout-changing copy of an fp32 parameter view into a contiguous <= 4-D tensor: one launch
    # (csrc/reduce.hip multi_permute_cast_kernel) instead of one strided-copy kernel per tensor
    rest = [i for i in range(len(dst)) if i not in skip]
    pm = [i for i in rest if dst[i].is_cuda and dst[i].is_contiguous() and dst[i].dtype in _PERM_DT
          and src[i].dtype == torch.float32 and 1 <= dst[i].dim() <= 4 and dst[i].numel() > 0
          and src[i].shape == dst[i].shape]
    permutes = []
    if len(pm) >= 4:
        by_dev = {}
        for i in pm:
            by_dev.setdefault(dst[i].device, []).append(i)
        for dev, idx in by_dev.items():
            permutes.append(_permute_table([dst[i] for i in idx], [src[i] for i in idx], dev) +
                            (dev, [dst[i] for i in idx], [src[i] for i in idx]))
        skip = skip | set(pm)
    return {"entries": entries, "transposes": transposes, "permutes": permutes,
            # (the tensors are kept next to the table: they own the memory the table points into)
            "casts": [_cast_table([dst[i] for i in idx], [src[i] for i in idx], dev) + (dev, [dst[i] for i in idx],
                                                                                         [src[i] for i in idx])
                      for dev, idx in casts.items()],
            "rest": ([d for i, d in enumerate(dst) if i not in skip], [s_ for i, s_ in enumerate(src) if i not in skip]),
            "gen": _GENERATION[0]}


_PERM_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _permute_table(dst, src, dev):
    """Chunk table of the multi-tensor layout-changing copy: 12 int64 per chunk (csrc/reduce.hip PermChunk), built once."""
    import numpy as np
    chunk = _lib.load_library().rfn_multi_permute_chunk_elems()
    rows = []
    for d, s_ in zip(dst, src):
        shape = [1] * (4 - d.dim()) + list(d.shape)
        strides = [0] * (4 - d.dim()) + list(s_.stride())
        n = d.numel()
        assert n < 2 ** 31, "the permute kernel indexes one tensor with 32 bits"
        rows += [(s_.data_ptr(), d.data_ptr(), shape[1], shape[2], shape[3], strides[0], strides[1], strides[2], strides[3],
                  off, min(chunk, n - off), _PERM_DT[d.dtype]) for off in range(0, n, chunk)]
    return torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev), len(rows)


def _cast_table(dst, src, dev):
    """Chunk table of the multi-tensor fp32 -> bf16 cast kernel: {src*, dst*, n}, 24 bytes per chunk, built once."""
    import numpy as np
    chunk = _lib.load_library().rfn_multi_cast_chunk_elems()
    rows = []
    for s_, d in zip(src, dst):
        n, sp, dp = s_.numel(), s_.data_ptr(), d.data_ptr()
        rows += [(sp + 4 * off, dp + 2 * off, min(chunk, n - off)) for off in range(0, n, chunk)]
    return torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev), len(rows)


def _transpose_table(dst, src, dev):
    """Tile table of the multi-tensor transpose + cast kernel: src views are (K, N) transposes of contiguous (N, K)
    fp32 parameters, dst the contiguous (K, N) bf16 copies."""
    import numpy as np
    rows = []
    T = _lib.load_library().rfn_multi_transpose_tile()
    for s_, d in zip(src, dst):
        K, N = s_.shape                                   # view (K, N) of a parameter stored (N, K)
        sp, dp = s_.data_ptr(), d.data_ptr()
        for n0 in range(0, N, T):
            for k0 in range(0, K, T):
                rows.append((sp, dp, N | (K << 32), n0 | (k0 << 32)))
    return torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev), len(rows)


def _multi_cast(table, nrows, dev):
    with on_device(dev):
        rc = _lib.load_library().rfn_multi_cast_f32_bf16(ptr(table), nrows, current_stream(dev))
    _lib.check(rc, "multi_cast_f32_bf16")


def _optimizer_step_post_hook(optimizer, args, kwargs):
    refresh((p for g in optimizer.param_groups for p in g["params"]), plan_key=("optimizer", id(optimizer)))


register_optimizer_step_post_hook(_optimizer_step_post_hook)


def refresh(params, plan_key=None):
    """Bring the cached copies of `params` up to date after the parameters were updated in place by something that
    does not bump their version counters (fused / foreach optimizers, EMA updates through `.data`): copies with a
    refill view are re-filled with ONE multi-tensor cast launch + one multi-tensor copy, every other derived tensor is
    dropped and re-made on next use.  Called by an optimizer-step post hook (every torch optimizer) and by the EMA
    update.  `plan_key`: callers that refresh the SAME parameter set every step name it; the (dst, src) lists are then
    built once (walking 1 000 parameters and building their views costs ~3 ms of host time per call) and reused until
    a cache entry is created / dropped or a parameter moves."""
    plan = _PLANS.get(plan_key) if plan_key is not None else None
    if plan is not None and plan["gen"] == _GENERATION[0]:
        ents = plan["entries"]
        # the parameters of a set are updated (and moved) together: the first and the last entry tell whether the
        # version stamps need a pass and whether the storage is still where the plan's views point
        probe = [(c.get(k), p) for c, k, p in ((ents[0], ents[-1]) if ents else ())]
        if any(e is None or e[2] != p.data_ptr() for e, p in probe):
            plan = None                                # parameter re-allocated or entry gone: rebuild
        elif any(e[1] != p._version for e, p in probe):
            for cache, key, p in ents:
                ent = cache.get(key)
                if ent is not None and ent[1] != p._version:
                    cache[key] = (ent[0], p._version, ent[2], ent[3])
    else:
        plan = None
    if plan is None:
        plan = _build_plan(list(params))
        if plan_key is not None:
            if len(_PLANS) >= 8:
                _PLANS.clear()
            _PLANS[plan_key] = plan
    with torch.no_grad():
        for table, nrows, dev, _, _ in plan["casts"]:
            _multi_cast(table, nrows, dev)
        for table, ntiles, dev, _, _ in plan["transposes"]:
            with on_device(dev):
                rc = _lib.load_library().rfn_multi_transpose_cast_f32_bf16(ptr(table), ntiles, current_stream(dev))
            _lib.check(rc, "multi_transpose_cast")
        for table, nrows, dev, _, _ in plan["permutes"]:
            with on_device(dev):
                rc = _lib.load_library().rfn_multi_permute_cast_f32(ptr(table), nrows, current_stream(dev))
            _lib.check(rc, "multi_permute_cast")
        if plan["rest"][0]:
            torch._foreach_copy_(*plan["rest"])


_EMA_TABLES = {}


def ema_update(ema_params, live_params, momentum, plan_key):
    """ema <- momentum * ema + (1 - momentum) * live over two aligned parameter lists (segmentation_model.py:676-689) as
    ONE kernel launch (csrc/reduce.hip: rfn_multi_ema_f32) from a chunk table that is built once per parameter set,
    then refresh() of the teacher's cached copies.  Falls back to two multi-tensor torch ops off the GPU."""
    ema_params, live_params = list(ema_params), list(live_params)
    if not ema_params:
        return
    dev = ema_params[0].device
    if dev.type == "cuda" and all(p.dtype == torch.float32 and p.is_contiguous() for p in ema_params + live_params):
        sig = (len(ema_params), ema_params[0].data_ptr(), ema_params[-1].data_ptr(), live_params[0].data_ptr(),
               live_params[-1].data_ptr())
        ent = _EMA_TABLES.get(plan_key)
        lib = _lib.load_library()
        if ent is None or ent[0] != sig:
            import numpy as np
            chunk = lib.rfn_multi_cast_chunk_elems()
            rows = []
            for e, l_ in zip(ema_params, live_params):
                n, ep, lp = e.numel(), e.data_ptr(), l_.data_ptr()
                rows += [(ep + 4 * off, lp + 4 * off, 0, min(chunk, n - off)) for off in range(0, n, chunk)]
            table = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)
            ent = _EMA_TABLES[plan_key] = (sig, table, len(rows))
        with on_device(dev):
            rc = lib.rfn_multi_ema_f32(ptr(ent[1]), ent[2], float(momentum), current_stream(dev))
        _lib.check(rc, "multi_ema_f32")
    else:
        ema, live = [p.data for p in ema_params], [p.data for p in live_params]
        torch._foreach_mul_(ema, momentum)
        torch._foreach_add_(ema, live, alpha=1.0 - momentum)
    refresh(ema_params, plan_key=plan_key)


def as_dtype(p, dtype):
    """`p` in the compute dtype (None stays None): the parameter itself if it already has it, else the cached copy."""
    if p is None or p.dtype == dtype:
        return p
    return derived(p, dtype, lambda t: t.to(dtype), _identity)


def _t_view(t):
    return t.reshape(t.shape[0], -1).t()              # (N, K) or (N, K, 1, 1) -> (K, N) view


def transposed(p, dtype):
    """Contiguous transposed copy W^T (K, N) of a Linear weight (N, K) in the compute dtype: the B operand of the
    input-gradient GEMM dx = dy W run as an NT product (csrc/mfma_gemm.hip).  Cached like the plain 16-bit copy and
    re-filled in place after optimizer / EMA updates."""
    return derived(p, ("T", dtype), lambda t: t.reshape(t.shape[0], -1).to(dtype).t().contiguous(), _t_view)


def compute_dtype(x):
    """dtype the dense ops run in: the autocast dtype inside an autocast region, else x's."""
    if x.is_cuda and torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return x.dtype


def mark_grad_sink(p):
    """FlatGradBuffer: `p.grad` is a persistent, contiguous fp32 view that kernels may accumulate into directly."""
    p._rfn_grad_sink = True


def grad_sink(p):
    if p is None or not getattr(p, "_rfn_grad_sink", False):
        return None
    g = p.grad                 # looked up every time: zero_grad(set_to_none=True) or a foreign p.grad must be seen
    if g is None or g.dtype != torch.float32 or not g.is_cuda:
        return None
    return g


def sum_rows(x, out=None, accumulate=False):
    """out[n] (+)= sum over the leading dim of a contiguous (S, n) fp32/bf16 matrix, fp32 result (csrc/reduce.hip)."""
    S, n = x.shape
    if not x.is_cuda or n % 8 != 0 or x.dtype not in _DT or not x.is_contiguous():
        r = x.sum(0, dtype=torch.float32)
        if out is None:
            return r
        return out.add_(r) if accumulate else out.copy_(r)
    if out is None:
        out, accumulate = torch.empty(n, dtype=torch.float32, device=x.device), False
    lib = _lib.load_library()
    nb = lib.rfn_sum_rows_workspace_bytes(S, n)
    ws = workspace(nb, x.device)
    with on_device(x.device):
        rc = lib.rfn_sum_rows(ptr(x), ptr(out), ptr(ws), S, n, _DT[x.dtype], 1 if accumulate else 0,
                              current_stream(x.device))
    _lib.check(rc, "sum_rows")
    return out


def linear_param_grads(g2, part, gb_out, gw_out):
    """One Linear's parameter gradients, accumulated in place: gb_out (N) += column sum of g2 (T, N);
    gw_out (N*K) += sum over the leading dim of part (S, N*K).  Two launches (csrc/reduce.hip).  Returns False when
    the shapes are outside the fused kernel's domain (caller falls back to two sum_rows calls)."""
    T, N = g2.shape
    S, NK = part.shape
    if not (g2.is_cuda and g2.dtype == part.dtype and g2.dtype in _DT and T > 64 and S <= 64 and N % 8 == 0
            and NK % 8 == 0 and g2.is_contiguous() and part.is_contiguous()):
        return False
    lib = _lib.load_library()
    ws = workspace(lib.rfn_sum_rows_workspace_bytes(T, N), g2.device)
    with on_device(g2.device):
        rc = lib.rfn_linear_param_grads(ptr(g2), ptr(gb_out), ptr(ws), T, N, 1, ptr(part), ptr(gw_out), S, NK, 1,
                                        _DT[g2.dtype], current_stream(g2.device))
    _lib.check(rc, "linear_param_grads")
    return True
