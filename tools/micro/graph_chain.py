#!/usr/bin/env python3
"""tools/micro/graph_chain.py -- what ONE more kernel node costs in a replayed linear hipGraph: chains of N dependent tiny
kernels (4 KB zero-fill through the C ABI; a 4 KB ATen add_), time per node = launch gap + minimal kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from refign_amd import _lib  # noqa: E402
from refign_amd._tensor import current_stream, ptr  # noqa: E402

dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
big = torch.zeros(64 << 20, device=dev)
lib = _lib.load_library()


def chain(fn, n):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 5


for name, fn in (("aten add_ 4 KB", lambda: x.add_(1.0)),
                 ("aten add_ 256 MB", lambda: big.add_(1.0))):
    t1, t2 = chain(fn, 200), chain(fn, 1000)
    print(f"{name:22s} 200 nodes {t1:9.1f} us   1000 nodes {t2:9.1f} us   per node {(t2 - t1) / 800:6.2f} us", flush=True)
