#!/usr/bin/env python3
"""tools/ffn_bench.py -- the Mix-FFN front half (fc1 -> depthwise 3x3 -> GELU) of the EMA teacher's 40 views per MiT-B5 stage: the
fused kernel (csrc/mixffn.hip) against fc1 GEMM + depthwise/GELU kernel, HIP events, median of 20."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refign_amd import dwconv  # noqa: E402
from refign_amd.seg import Mlp  # noqa: E402

dev = torch.device("cuda:0")
views = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


for stage, (H, W, C) in enumerate([(135, 240, 64), (68, 120, 128), (34, 60, 320), (17, 30, 512)], 1):
    torch.manual_seed(stage)
    mlp = Mlp(C, 4 * C).to(dev).eval()
    x = torch.randn(views, H * W, C, device=dev).to(torch.bfloat16)
    dw = mlp.dwconv.dwconv
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        fused = timed(lambda: dwconv.ffn_fc1_dw_gelu(x, mlp.fc1, dw, H, W))
        h = mlp.fc1(x)
        t1 = timed(lambda: mlp.fc1(x))
        t2 = timed(lambda: dwconv.dwconv3x3_gelu_tokens(h, dw.weight, dw.bias, H, W))
    gf = 2.0 * views * H * W * C * 4 * C / 1e9
    print(f"stage {stage}: {views} x {H}x{W} tokens, C {C:3d}: fused {fused:7.1f} us | fc1 {t1:7.1f} + dw/gelu {t2:7.1f} = {t1 + t2:7.1f} us | "
          f"fc1 {gf:6.1f} GF, hidden tensor {views * H * W * 4 * C * 2 / 1e6:6.1f} MB", flush=True)
