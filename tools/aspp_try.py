#!/usr/bin/env python3
"""tools/aspp_try.py -- the EMA teacher's three dilated depthwise ASPP branches (40 x 135 x 240 x 1024 bf16, dilations 6 / 12 /
18; daformer.py:46-62): statistics pass + fused BatchNorm/ReLU pass, as the step runs them (whole batch, one dilation after
the other) and image by image (all three dilations on one 66 MB map while it is in the Infinity Cache)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refign_amd import _lib  # noqa: E402
from refign_amd._tensor import current_stream, on_device, ptr  # noqa: E402

dev = torch.device("cuda:0")
B, H, W, C = 40, 135, 240, 1024
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, C, generator=g).to(dev).to(torch.bfloat16)
w = [torch.randn(9, C, generator=g).to(dev) for _ in range(3)]
b = [torch.randn(C, generator=g).to(dev) for _ in range(3)]
sums = [torch.zeros(2 * C + 1, dtype=torch.float64, device=dev) for _ in range(3)]
ys = [torch.empty_like(x) for _ in range(3)]
gam = torch.ones(C, device=dev)
bet = torch.zeros(C, device=dev)
lib = _lib.load_library()
dil = (6, 12, 18)


def stats(xs, k):
    rc = lib.rfn_dwconv3x3_nhwc_stats(ptr(xs), ptr(w[k]), ptr(b[k]), ptr(sums[k]), xs.shape[0], H, W, C, dil[k], 1, current_stream(dev))
    assert rc == 0


def apply(xs, ysl, k):
    rc = lib.rfn_dwconv3x3_bn_act_nhwc_fwd(ptr(xs), ptr(w[k]), ptr(b[k]), ptr(gam), ptr(bet), ptr(sums[k]), None, None, ptr(ysl),
                                           xs.shape[0], H, W, C, dil[k], 1e-5, 0.1, 1, 1, current_stream(dev))
    assert rc == 0


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def whole():
    for k in range(3):
        stats(x, k)
    for k in range(3):
        apply(x, ys[k], k)


def per_image(n):
    def f():
        for i in range(0, B, n):
            for k in range(3):
                stats(x[i:i + n], k)
        for i in range(0, B, n):
            for k in range(3):
                apply(x[i:i + n], ys[k][i:i + n], k)
    return f


def tri():
    import ctypes
    w3 = torch.stack(w).contiguous(); b3 = torch.stack(b).contiguous()
    s3 = torch.zeros(3, 2 * C + 1, dtype=torch.float64, device=dev)
    rc = lib.rfn_dwconv3x3_tri_stats(ptr(x), ptr(w3), ptr(b3), ptr(s3), B, H, W, C, 6, current_stream(dev)); assert rc == 0
    arr = lambda ts: (ctypes.c_void_p * 3)(*[ptr(t) for t in ts])
    rc = lib.rfn_dwconv3x3_tri_bn_act_fwd(ptr(x), ptr(w3), ptr(b3), arr([gam] * 3), arr([bet] * 3), ptr(s3), arr([None] * 3), arr([None] * 3),
                                          arr(ys), B, H, W, C, 6, (ctypes.c_float * 3)(1e-5, 1e-5, 1e-5), (ctypes.c_float * 3)(0.1, 0.1, 0.1), 1,
                                          current_stream(dev)); assert rc == 0


def tri_stats():
    w3 = torch.stack(w).contiguous(); b3 = torch.stack(b).contiguous()
    s3 = torch.zeros(3, 2 * C + 1, dtype=torch.float64, device=dev)
    rc = lib.rfn_dwconv3x3_tri_stats(ptr(x), ptr(w3), ptr(b3), ptr(s3), B, H, W, C, 6, current_stream(dev)); assert rc == 0


with on_device(dev):
    print(f"three dilations in one pass each (tri kernel): {timeit(tri):.3f} ms  (stats only: {timeit(tri_stats):.3f} ms)")
    if os.environ.get("RFN_TRI_ONLY"):
        sys.exit(0)
    print(f"whole batch, dilation after dilation: {timeit(whole):.3f} ms  (stats only: {timeit(lambda: [stats(x, k) for k in range(3)]):.3f} ms)")
    for n in (1, 2, 4, 8):
        print(f"{n} image(s) at a time, three dilations each: {timeit(per_image(n)):.3f} ms")
