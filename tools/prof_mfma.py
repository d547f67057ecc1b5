#!/usr/bin/env python3
"""tools/prof_mfma.py -- run ONE matrix-core kernel of the step a few times (for rocprofv3 --kernel-trace / --pmc).
usage: prof_mfma.py [gemm_fc1_s3 | gemm_fc2_s3 | gemm_fc1_s1 | f8gemm_fc1_s3 | f8gemm_fc2_s3 | f8attn_s3 | conv_bottleneck |
                     attn_fwd_s3 | attn_bwd_s3 | wgrad_s3] [reps]
Shapes: MiT-B5 stage shapes of the HRDA step (teacher batch 40 x 540x960 views; student batch 4)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refign_amd import mfma  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "gemm_fc1_s3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
dt = torch.bfloat16
r = lambda *s: torch.randn(*s, device=dev).to(dt)  # noqa: E731
if what.startswith("gemm_"):
    M, N, K = {"gemm_fc1_s3": (81600, 1280, 320), "gemm_fc2_s3": (81600, 320, 1280), "gemm_fc1_s1": (1296000, 256, 64)}[what]
    x, w, b = r(M, K), r(N, K), r(N)
    fn = lambda: mfma.gemm_nt(x, w, b)  # noqa: E731
    print(f"{what}: M={M} N={N} K={K} flops={2.0 * M * N * K:.4g} bytes={2.0 * (M * K + N * K + M * N):.4g}")
elif what.startswith("f8gemm_"):
    from refign_amd import f8
    M, N, K = {"f8gemm_fc1_s3": (81600, 1280, 320), "f8gemm_fc2_s3": (81600, 320, 1280)}[what]
    x8, w8 = f8.quantize(r(M, K)), f8.quantize(r(N, K))
    ws, b = torch.ones(N, device=dev), r(N)
    fn = lambda: f8.gemm_nt(x8, w8, ws, bias=b, out_f8=what == "f8gemm_fc1_s3")  # noqa: E731
    print(f"{what}: M={M} N={N} K={K} flops={2.0 * M * N * K:.4g} bytes={1.0 * (M * K + N * K) + (1.0 if 'fc1' in what else 2.0) * M * N:.4g}")
elif what == "f8attn_s3":
    from refign_amd import f8
    B, h, Nq, Nkv = 40, 5, 2040, 510
    q8, kv8 = f8.quantize(r(B, Nq, h * 64)), f8.quantize(r(B, Nkv, 2 * h * 64))
    fn = lambda: f8.attention(q8, kv8, h, 0.125)  # noqa: E731
    print(f"{what}: B={B} heads={h} Nq={Nq} Nkv={Nkv} fwd flops={4.0 * B * h * Nq * Nkv * 64:.4g}")
elif what.startswith("ffn_"):
    from refign_amd import dwconv
    from refign_amd.seg import Mlp
    H, W, C = {"ffn_s1": (135, 240, 64), "ffn_s2": (68, 120, 128), "ffn_s3": (34, 60, 320)}[what]
    mlp = Mlp(C, 4 * C).to(dev).eval()
    x = r(40, H * W, C)
    print(f"{what}: 40 x {H}x{W} tokens, C={C}: fc1 flops={2.0 * 40 * H * W * C * 4 * C:.4g} hidden bytes={2.0 * 40 * H * W * 4 * C:.4g}")

    def fn():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return dwconv.ffn_fc1_dw_gelu(x, mlp.fc1, mlp.dwconv.dwconv, H, W)
elif what == "wgrad_s3":
    T, N, K = 8160, 1280, 320
    g, x = r(T, N), r(T, K)
    gw, gb = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    fn = lambda: mfma.gemm_tn(g, x, out=gw, bias_out=gb)  # noqa: E731
    print(f"{what}: T={T} N={N} K={K} flops={2.0 * T * N * K:.4g}")
elif what == "conv_bottleneck":
    B, H, W, C, N = 40, 135, 240, 1024, 256
    x = r(B, H, W, C)
    wp = mfma.pack_conv_weight(r(N, C, 3, 3), dt)
    fn = lambda: mfma.conv2d_nhwc(x, wp, None, 3, 3, 1, 1, 1)  # noqa: E731
    print(f"{what}: {B}x{H}x{W}x{C} -> {N}, 3x3: flops={2.0 * B * H * W * N * 9 * C:.4g}")
else:
    B, h, Nq, Nkv = (40, 5, 2040, 510) if what == "attn_fwd_s3" else (4, 5, 2040, 510)
    q = r(B, Nq, h * 64).requires_grad_(what != "attn_fwd_s3")
    kv = r(B, Nkv, 2 * h * 64).requires_grad_(what != "attn_fwd_s3")
    go = r(B, Nq, h * 64)
    print(f"{what}: B={B} heads={h} Nq={Nq} Nkv={Nkv} fwd flops={4.0 * B * h * Nq * Nkv * 64:.4g}")
    if what == "attn_fwd_s3":
        def fn():
            with torch.no_grad():
                return mfma.attention(q, kv, h, 0.125)
    else:
        def fn():
            o = mfma.attention(q, kv, h, 0.125)
            return torch.autograd.grad(o, (q, kv), go)
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print("done", what, reps)
