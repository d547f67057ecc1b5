#!/usr/bin/env python3
"""tools/attn_bench.py -- the attention kernels (csrc/attn.hip) at the shapes of the Refign HRDA step, timed from a
replayed hipGraph of 10 back-to-back calls (no host launch gaps): forward, and forward + backward (dQ, pack, dK/dV,
finish) of the student's shapes; forward only of the teacher's.  TFLOP/s: 4 B h Nq Nkv 64 forward, 2.5 x backward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refign_amd import mfma  # noqa: E402


def graph_time(fn, inner=10, reps=10):
    for _ in range(3):
        fn()
    cur, side = torch.cuda.current_stream(), torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):                        # stream-local workspaces exist before the capture
        fn()
    cur.wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        keep = [fn() for _ in range(inner)]
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    del keep
    return e0.elapsed_time(e1) * 1e3 / (reps * inner)


def main():
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    print("# B heads Nq Nkv            fwd us (TF/s)     bwd us (TF/s)")
    for B in (4, 40):
        for st, h, Nq, Nkv in (("s1", 1, 32400, 480), ("s2", 2, 8160, 510), ("s3", 5, 2040, 510), ("s4", 8, 510, 510)):
            C = h * 64
            q = torch.randn(B, Nq, C, device=dev).to(dt)
            kv = torch.randn(B, Nkv, 2 * C, device=dev).to(dt)
            go = torch.randn(B, Nq, C, device=dev).to(dt)
            fl = 4.0 * B * h * Nq * Nkv * 64
            with torch.no_grad():
                t0 = graph_time(lambda: mfma.attention(q, kv, h, 0.125))
            line = f"B={B:2d} {st} h={h} {Nq:6d} x {Nkv:4d}   {t0:8.1f} ({fl / t0 / 1e6:6.1f})"
            if B == 4:
                qg, kvg = q.clone().requires_grad_(), kv.clone().requires_grad_()
                # forward + backward in one captured call (the autograd engine runs a node on the stream of its forward);
                # the forward with the packs a backward needs is timed on its own and subtracted
                t0g = graph_time(lambda: mfma.attention(qg, kvg, h, 0.125))
                t1 = graph_time(lambda: torch.autograd.grad(mfma.attention(qg, kvg, h, 0.125), (qg, kvg), go)) - t0g
                line += f"   {t1:8.1f} ({2.5 * fl / t1 / 1e6:6.1f})"
            print(line, flush=True)


if __name__ == "__main__":
    main()
