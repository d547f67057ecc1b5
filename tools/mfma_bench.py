#!/usr/bin/env python3
"""tools/mfma_bench.py -- the hand-written matrix-core kernels (csrc/mfma_gemm.hip, csrc/attn.hip) at the shapes of the
Refign HRDA step (MiT-B5, 540x960 views: student batch 4, teacher batch 40), HIP events on the launch stream, beside the
ROCm-library kernel torch dispatches to for the same op (hipBLASLt / fused SDPA).  TFLOP/s = 2 M N K (GEMM),
4 B h Nq Nkv 64 (attention forward), 2.5 x that (backward: 5 GEMMs)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refign_amd import mfma  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    print("== GEMM (bf16)  T x K -> N      ours us (TF/s)   library us (TF/s)")
    tok = {"s1": 32400, "s2": 8160, "s3": 2040, "s4": 510}
    dims = {"s1": 64, "s2": 128, "s3": 320, "s4": 512}
    for B in (4, 40):
        for st in ("s1", "s2", "s3", "s4"):
            T, C = B * tok[st], dims[st]
            for name, N, K in (("q/proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)):
                x = torch.randn(T, K, device=dev).to(dt)
                w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
                b = torch.randn(N, device=dev).to(dt)
                fl = 2.0 * T * N * K
                t0 = timeit(lambda: mfma.gemm_nt(x, w, b))
                t1 = timeit(lambda: F.linear(x, w, b))
                line = f"B={B:2d} {st} {name:7s} fwd   {T:8d} x {K:5d} -> {N:5d}   {t0:8.1f} ({fl / t0 / 1e6:6.1f})   {t1:8.1f} ({fl / t1 / 1e6:6.1f})"
                if B == 4:
                    g = torch.randn(T, N, device=dev).to(dt)
                    t2 = timeit(lambda: mfma.gemm_tn(g, x))
                    t3 = timeit(lambda: g.t().mm(x))
                    line += f"   wgrad {t2:8.1f} ({fl / t2 / 1e6:6.1f})   {t3:8.1f} ({fl / t3 / 1e6:6.1f})"
                print(line, flush=True)
    print("== attention (bf16, d=64)   ours fwd / bwd us (TF/s)    SDPA fwd / bwd us (TF/s)")
    for B in (4, 40):
        for st, h, Nq, Nkv in (("s1", 1, 32400, 480), ("s2", 2, 8160, 510), ("s3", 5, 2040, 510), ("s4", 8, 510, 510)):
            C = h * 64
            q = torch.randn(B, Nq, C, device=dev).to(dt).requires_grad_(B == 4)
            kv = torch.randn(B, Nkv, 2 * C, device=dev).to(dt).requires_grad_(B == 4)
            go = torch.randn(B, Nq, C, device=dev).to(dt)
            fl = 4.0 * B * h * Nq * Nkv * 64

            def sdpa():
                qh = q.view(B, Nq, h, 64).transpose(1, 2)
                k, v = kv.view(B, Nkv, 2, h, 64).permute(2, 0, 3, 1, 4).unbind(0)
                return F.scaled_dot_product_attention(qh, k, v, scale=0.125).transpose(1, 2).reshape(B, Nq, C)

            if B == 4:
                t0 = timeit(lambda: mfma.attention(q, kv, h, 0.125))
                t1 = timeit(lambda: sdpa())
                o = mfma.attention(q, kv, h, 0.125)
                tb0 = timeit(lambda: torch.autograd.grad(o, (q, kv), go, retain_graph=True))
                o2 = sdpa()
                tb1 = timeit(lambda: torch.autograd.grad(o2, (q, kv), go, retain_graph=True))
                print(f"B={B:2d} {st} h={h} Nq={Nq:6d} Nkv={Nkv:4d}  {t0:8.1f} ({fl / t0 / 1e6:6.1f}) / {tb0:8.1f} ({2.5 * fl / tb0 / 1e6:6.1f})"
                      f"    {t1:8.1f} ({fl / t1 / 1e6:6.1f}) / {tb1:8.1f} ({2.5 * fl / tb1 / 1e6:6.1f})", flush=True)
            else:
                with torch.no_grad():
                    t0 = timeit(lambda: mfma.attention(q, kv, h, 0.125))
                    t1 = timeit(lambda: sdpa())
                print(f"B={B:2d} {st} h={h} Nq={Nq:6d} Nkv={Nkv:4d}  {t0:8.1f} ({fl / t0 / 1e6:6.1f})    {t1:8.1f} ({fl / t1 / 1e6:6.1f})", flush=True)


if __name__ == "__main__":
    main()
