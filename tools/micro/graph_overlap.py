#!/usr/bin/env python3
"""tools/micro/graph_overlap.py -- do two streams overlap when one or both of them replay a hipGraph?
Small kernels (a 512^2 matmul chain fills a fraction of the chip), so perfect overlap ~ max(A, B), none ~ A + B."""
import time

import torch

dev = torch.device("cuda:0")
N, LEN = 512, 1500


def chain(x, w, n=LEN):
    for _ in range(n):
        x = torch.mm(x, w)
    return x


def make():
    return torch.randn(N, N, device=dev) * 0.01, torch.eye(N, device=dev)


def graph_of(x, w):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain(x, w, 3)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = chain(x, w)
    return g, y


xa, wa = make()
xb, wb = make()
ga, _ = graph_of(xa, wa)
gb, _ = graph_of(xb, wb)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)


def run(a, b, main_default=False):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st_a = torch.cuda.current_stream() if main_default else sa
    if a:
        with torch.cuda.stream(st_a):
            ga.replay() if a == "graph" else chain(xa, wa)
    if b:
        with torch.cuda.stream(sb):
            gb.replay() if b == "graph" else chain(xb, wb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(2):
    for a, b, d in [("eager", None, False), ("graph", None, False), (None, "eager", False), (None, "graph", False),
                    ("eager", "eager", False), ("graph", "eager", False), ("eager", "graph", False),
                    ("graph", "graph", False), ("graph", "eager", True), ("graph", "graph", True)]:
        print(f"A={str(a):6s} B={str(b):6s} A on {'default' if d else 'pool   '} stream: {run(a, b, d):7.2f} ms")
    print()


def run_event(a, b, flush):
    """B waits for an event recorded on A's stream BEFORE A's work is queued (the step's pattern)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        ev = sa.record_event()
        if flush:
            ev.query()
        ga.replay() if a == "graph" else chain(xa, wa)
    with torch.cuda.stream(sb):
        sb.wait_event(ev)
        gb.replay() if b == "graph" else chain(xb, wb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for a, b in [("graph", "graph"), ("graph", "eager"), ("eager", "graph")]:
    for flush in (False, True):
        print(f"event before A; A={a:6s} B={b:6s} query={flush}: {run_event(a, b, flush):7.2f} ms")
