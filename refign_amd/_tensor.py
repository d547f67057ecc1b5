"""Tensor plumbing between torch (device memory + streams) and the C ABI.  torch is plumbing only."""
import torch


def require_device_tensor(t, name, dtype=None):
    """Mirror of the reference's CHECK_CUDA / CHECK_CONTIGUOUS (correlation_sampler.cpp:13-16): RuntimeError."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a HIP (cuda:N) tensor: refign_amd has no CPU path")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must have dtype {dtype}, got {t.dtype}")
    return t


def same_device(*ts):
    dev = ts[0].device
    for t in ts[1:]:
        if t is not None and t.device != dev:
            raise RuntimeError("all tensors must be on the same device")  # CHECK_SAME_DEVICE
    return dev


def ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = torch._C._cuda_getCurrentRawStream


def current_stream(device):
    """Raw hipStream_t of torch's current stream on `device` (the reference launches on the legacy default
    stream, correlation_cuda_kernel.cu:271; we honour torch's stream semantics instead so graphs/streams work).
    Straight from the C binding: this sits on the launch path of ~3 000 kernel calls per training step."""
    return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """`with on_device(dev):` == `with torch.cuda.device(dev):` without the guard's cost when `dev` already is the
    current device (always, with one process per GPU)."""
    if device.index is None or torch.cuda.current_device() == device.index:
        return _NO_GUARD
    return torch.cuda.device(device)


_WS = {}


def workspace(nbytes, device):
    """Scratch buffer for kernels that run back to back on one stream: a cached per-device allocation that only grows,
    instead of a caching-allocator round trip per call.  NOT for results, and not while a graph is being captured
    (captured kernels keep their pointers: they get a private allocation)."""
    if nbytes <= 0:
        return None
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    key = (device, _raw_stream(device.index if device.index is not None else torch.cuda.current_device()))
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _WS[key] = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8, device=device)
    return buf


_CONSTS = {}


def const_tensor(values, like, dtype=None):
    """A small constant tensor on `like`'s device, created ONCE per (values, dtype, device): `like.new_tensor([...])`
    inside the step is a pageable host-to-device copy, i.e. a host synchronisation that drains the launch queue."""
    dtype = dtype or like.dtype
    flat = tuple(torch.as_tensor(values, dtype=torch.float64).flatten().tolist())
    shape = tuple(torch.as_tensor(values).shape)
    key = (flat, shape, dtype, like.device)
    t = _CONSTS.get(key)
    if t is None:
        if len(_CONSTS) > 4096:
            _CONSTS.clear()
        t = _CONSTS[key] = torch.tensor(values, dtype=dtype, device=like.device)
    return t


def upload_async(values, dtype, device):
    """Per-step host data (e.g. random class choices) to the device without a host synchronisation: staged through
    pinned memory, copy enqueued on the current stream."""
    t = torch.as_tensor(values, dtype=dtype)
    if device.type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)
