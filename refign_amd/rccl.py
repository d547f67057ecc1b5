"""RCCL called directly (ctypes on the librccl.so that torch ships and has already loaded): an all-reduce that is ONE
kernel on the caller's stream.

Why: torch's process group runs every collective on a stream of its own and joins it to the caller's stream with events.
Inside a hipGraph capture each SyncBatchNorm statistics exchange of a student pass therefore becomes a cross-stream
branch of the graph, hipGraph replays such branches with a synchronisation per edge, and two passes replaying next to
each other pay for it (a communicator per pass through torch's process group: 237.9 ms/step against 222.5 ms with the
passes in stream order -- DESIGN.md section 6).  `ncclAllReduce(..., comm, stream)` on the capture stream is a plain
kernel node: the captured pass stays a linear graph.

`DirectComm(group)`: rank 0 of `group` makes a ncclUniqueId, torch's process group broadcasts its 128 bytes, every rank
calls ncclCommInitRank.  One DirectComm per stream that may run collectives concurrently (same rule as everywhere: two
streams must not issue collectives of one communicator in a rank-dependent order).

Used under data parallelism with RFN_DDP_MODE=direct / direct3 (bn.ddp_mode; the default `torch` sends every exchange
through torch's process group and keeps the student passes eager).  It could only be run with a 1-rank communicator on the one-GPU development boxes
(tests/test_syncbn_gpu.py; RFN_DDP_REHEARSAL): 192.3 ms/step for one rank of N with graphed student passes, against
216.4 ms with the same graphs over torch's process group and 214-231 ms with the eager student
(profiles/r02_ddp_rehearsal.txt).
"""
import ctypes
import os

import torch
import torch.distributed as dist

_NCCL_DTYPE, _NCCL_SUM = {torch.float32: 7, torch.float64: 8}, 0          # ncclDataType_t / ncclRedOp_t (nccl.h)


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


_lib = None
_LIVE = []          # communicators in creation order (the same on every rank)


def _rccl():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = ctypes.CDLL(path)
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclCommDestroy):
            f.restype = ctypes.c_int
        # ABI check: the enum values and the 128-byte id above are those of the NCCL 2.x API (nccl.h: ncclFloat32 = 7, ncclFloat64 = 8,
        # ncclSum = 0, NCCL_UNIQUE_ID_BYTES = 128, unchanged since 2.0); refuse anything else instead of guessing
        lib.ncclGetVersion.argtypes = [ctypes.POINTER(ctypes.c_int)]
        lib.ncclGetVersion.restype = ctypes.c_int
        v = ctypes.c_int(0)
        if lib.ncclGetVersion(ctypes.byref(v)) != 0 or not (20000 <= v.value < 30000 or 2000 <= v.value < 3000):
            raise RuntimeError(f"rccl: {path} reports version code {v.value}; this binding is written against the NCCL 2.x "
                               f"API (ncclGetVersion 2xxxx)")
        global VERSION
        VERSION = v.value
        _lib = lib
    return _lib


VERSION = None


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"rccl: {what}: {_rccl().ncclGetErrorString(rc).decode()}")


def enabled():
    from .bn import ddp_mode
    return ddp_mode() != "torch"


class DirectComm:
    """A communicator of our own over the ranks of `group` (default: the world) on `device`."""

    def __init__(self, device, group=None):
        lib = _rccl()
        self.device = torch.device(device)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        uid = _UniqueId()
        if self.rank == 0:
            _check(lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        raw = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=self.device)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(raw, src=src, group=group)
        ctypes.memmove(ctypes.byref(uid), bytes(raw.cpu().tolist()), 128)
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")
        _LIVE.append(self)

    def all_reduce_(self, t):
        """in-place sum of a contiguous float32 / float64 tensor over the ranks, on the CURRENT stream of its device"""
        if t.dtype not in _NCCL_DTYPE or not t.is_contiguous() or t.device != self.device:
            raise RuntimeError("rccl.DirectComm.all_reduce_: contiguous float32/float64 tensor on the communicator's device")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        with torch.cuda.device(self.device):
            _check(_rccl().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _NCCL_DTYPE[t.dtype], _NCCL_SUM, self._comm,
                                         stream), "ncclAllReduce")
        return t

    def destroy(self):
        if self._comm:
            _rccl().ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()
        if self in _LIVE:
            _LIVE.remove(self)


def destroy_all():
    """Every rank, same order, after the last collective has completed (call torch.cuda.synchronize() first) and before
    torch.distributed.destroy_process_group()."""
    for c in list(_LIVE):
        c.destroy()
