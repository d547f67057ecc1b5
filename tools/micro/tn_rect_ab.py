#!/usr/bin/env python3
"""tools/micro/tn_rect_ab.py -- weight-gradient GEMMs of MiT-B5's 320-wide stage (student: 8160 tokens) with 64 x 64 tiles
(RFN_GEMM_TN_RECT=0) against the rectangular 64 x 128 / 128 x 64 ones, accumulate mode as in the step, in a replayed graph."""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def child():
    from refign_amd.mfma import gemm_tn
    dev = torch.device("cuda:0")
    for (T, N, K, seg) in [(8160, 320, 1280, False), (8160, 1280, 320, False), (8160, 320, 1280, True), (8160, 1280, 320, True),
                           (130560, 64, 256, False), (130560, 256, 64, False), (81600, 320, 1280, False)]:
        g = torch.randn(T, N, device=dev).bfloat16()
        x = torch.randn(T, K, device=dev).bfloat16()
        gw, gb = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
        rs = torch.rand(4, device=dev) if seg else None
        kw = dict(rowscale=rs, rows_per_sample=T // 4) if seg else {}
        f = lambda: gemm_tn(g, x, out=gw, bias_out=gb, **kw)  # noqa: E731
        assert f() is not None
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                f()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 10
        print(f"  T={T:6d} N={N:4d} K={K:4d} {'rowscale' if seg else '        '}  {us:7.1f} us  {2e-6 * T * N * K / us:6.1f} TFLOP/s")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for rect in ("0", "1"):
            print(f"RFN_GEMM_TN_RECT={rect}", flush=True)
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, RFN_GEMM_TN_RECT=rect))
