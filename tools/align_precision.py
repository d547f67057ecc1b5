#!/usr/bin/env python3
"""tools/align_precision.py -- the align step at 1080x1920 with its convolutions in fp32 / fp16 / bf16 (HIP correlation,
warp, L2-norm and uncertainty kernels always fp32): time, and how far flow / confidence / warped logits move from the
fp32 result.  Images: a smooth random texture and a warped + noisy copy of it (so that there IS a flow to find)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from refign_amd import align as A  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.WORKLOADS["refign_hrda_step_1080x1920"](dev, 2, 1234, 1080, 1920, "bf16")
m = wl.model
g = torch.Generator().manual_seed(5)
H, W = 1080, 1920
base = F.interpolate(torch.randn(2, 3, H // 8, W // 8, generator=g), size=(H, W), mode="bicubic", align_corners=False)
base = base + 0.3 * F.interpolate(torch.randn(2, 3, H // 2, W // 2, generator=g), size=(H, W), mode="bilinear")
shift = torch.roll(base, shifts=(7, -11), dims=(2, 3)) + 0.05 * torch.randn(2, 3, H, W, generator=g)
trg, ref = base.to(dev), shift.to(dev)
logits = (3 * torch.randn(2, 19, H // 8, W // 8, generator=g)).to(dev)
logits = F.interpolate(logits, size=(H, W), mode="bilinear")          # smooth logits, like a network's
out = {}
for mode in ("fp32", "fp16", "bf16"):
    os.environ["RFN_ALIGN_DTYPE"] = mode
    for _ in range(2):
        r = m.align(logits, ref, trg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        r = m.align(logits, ref, trg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    with torch.no_grad():
        pyr = A.extract_pyramids(m.alignment_backbone, ref, trg) if mode == "fp32" else None
    out[mode] = [t.float() for t in r]
    print(f"{mode}: align {dt:7.2f} ms", flush=True)
w0, m0, c0 = out["fp32"]
for mode in ("fp16", "bf16"):
    w1, m1, c1 = out[mode]
    both = (m0 > 0) & (m1 > 0)
    print(f"{mode} vs fp32: warped logits |d| mean {float((w1 - w0).abs().mean()):.4e} max {float((w1 - w0).abs().max()):.3e} "
          f"(|logit| mean {float(w0.abs().mean()):.3f});  mask differs on {float((m0 != m1).float().mean()) * 100:.3f}% px;  "
          f"confidence |d| mean {float((c1 - c0).abs().mean()):.3e} max {float((c1 - c0).abs().max()):.3e}; "
          f"argmax of warped logits differs on {float((w1.argmax(1) != w0.argmax(1))[both].float().mean()) * 100:.3f}% px")
