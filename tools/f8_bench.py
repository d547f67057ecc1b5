"""K5 micro-benchmark: the fp8 (e4m3) kernels of the EMA-teacher path against the bf16 kernels of the default mode on the
teacher's own shapes (40 HRDA views of 540x960 per GPU: 1 296 000 / 321 600 / 81 600 / 20 400 tokens per MiT stage).
HIP events on the launch stream, median of `--iters` launches.  `python tools/f8_bench.py [--views 40]`."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refign_amd import f8, mfma  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=40)
    ap.add_argument("--iters", type=int, default=15)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    V = args.views
    stages = [(64, 135 * 240, 480, 1, 8), (128, 68 * 120, 510, 2, 4), (320, 34 * 60, 510, 5, 2), (512, 17 * 30, 510, 8, 1)]
    print(f"{'op':34s} {'M':>9s} {'N':>6s} {'K':>6s} {'bf16 us':>9s} {'f8 us':>9s} {'f8->f8 us':>9s} {'x':>6s} {'f8 TF/s':>8s}")
    tot = [0.0, 0.0]
    for C, ntok, nkv, heads, sr in stages:
        M = V * ntok
        gemms = [("q", M, C, C), ("kv", V * nkv, 2 * C, C), ("proj", M, C, C), ("fc1", M, 4 * C, C), ("fc2", M, C, 4 * C)]
        if sr > 1:
            gemms.append(("sr", V * nkv, C, sr * sr * C))
        for name, m, n, k in gemms:
            x = torch.randn(m, k, device=dev).to(torch.bfloat16)
            w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
            b = torch.zeros(n, device=dev, dtype=torch.bfloat16)
            x8, w8 = f8.quantize(x), f8.quantize(w)
            ws = torch.ones(n, device=dev)
            t16 = timed(lambda: mfma.gemm_nt(x, w, b), args.iters)
            t8 = timed(lambda: f8.gemm_nt(x8, w8, ws, bias=b), args.iters)
            t88 = timed(lambda: f8.gemm_nt(x8, w8, ws, bias=b, out_f8=True), args.iters)
            best = t88 if name in ("q", "kv", "fc1") else t8
            tot[0] += t16
            tot[1] += best
            print(f"{'gemm C=%d %s' % (C, name):34s} {m:9d} {n:6d} {k:6d} {t16:9.1f} {t8:9.1f} {t88:9.1f} {t16 / best:6.2f} "
                  f"{2.0 * m * n * k / best / 1e6:8.0f}")
            del x, w, x8, w8
        q = torch.randn(V, ntok, C, device=dev).to(torch.bfloat16)
        kv = torch.randn(V, nkv, 2 * C, device=dev).to(torch.bfloat16)
        q8, kv8 = f8.quantize(q), f8.quantize(kv)
        with torch.no_grad():
            t16 = timed(lambda: mfma.attention(q, kv, heads, 0.125), args.iters)
        t8 = timed(lambda: f8.attention(q8, kv8, heads, 0.125), args.iters)
        tot[0] += t16
        tot[1] += t8
        print(f"{'attention C=%d (pack + fwd)' % C:34s} {V * ntok:9d} {nkv:6d} {64:6d} {t16:9.1f} {t8:9.1f} {'':9s} {t16 / t8:6.2f} "
              f"{4.0 * V * heads * ntok * nkv * 64 / t8 / 1e6:8.0f}")
        del q, kv, q8, kv8
        torch.cuda.empty_cache()
    print(f"sum over one block of each stage: bf16 {tot[0]:.0f} us, fp8 {tot[1]:.0f} us ({tot[0] / tot[1]:.2f}x)")


if __name__ == "__main__":
    main()
