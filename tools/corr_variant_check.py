#!/usr/bin/env python3
"""tools/corr_variant_check.py -- correctness (against a plain torch formulation) and launch time of the patch-9
correlation forward for the kernel variant selected by RFN_CORR_VARIANT (one process per variant: the library reads
the variable once)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refign_amd import correlation  # noqa: E402


def ref_corr(f1, f2):
    b, c, h, w = f1.shape
    p = F.pad(f2.double(), (4, 4, 4, 4))
    out = torch.empty(b, 9, 9, h, w, dtype=torch.float64, device=f1.device)
    for dy in range(9):
        for dx in range(9):
            out[:, dy, dx] = (f1.double() * p[:, :, dy:dy + h, dx:dx + w]).sum(1)
    return out


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
    g = torch.Generator().manual_seed(1)
    print("variant", os.environ.get("RFN_CORR_VARIANT", "0"))
    worst = 0.0
    for (b, c, h, w) in [(1, 8, 16, 32), (2, 16, 37, 52), (1, 64, 70, 100), (2, 128, 33, 64), (3, 24, 5, 8), (1, 8, 16, 36)]:
        f1 = torch.randn(b, c, h, w, generator=g).to(dev)
        f2 = torch.randn(b, c, h, w, generator=g).to(dev)
        want = ref_corr(f1, f2)
        got = correlation.forward(f1, f2, 1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1).double()
        e = (got - want).abs().max().item() / want.abs().max().item()
        wn = F.normalize(F.relu(want.view(b, 81, h, w)), dim=1)
        gn = correlation.local_correlation_layer(f2, f1).double()
        en = (gn - wn).abs().max().item()
        worst = max(worst, e, en)
        print(f"  {b}x{c}x{h}x{w}: raw rel err {e:.2e}   fused abs err {en:.2e}")
    print("worst", worst, "OK" if worst < 1e-5 else "FAIL")
    if worst >= 1e-5:
        sys.exit(1)
    if os.environ.get("RFN_CORR_CHECK_ONLY"):
        return
    for (lvl, C, H, W) in [("L1", 128, 270, 480), ("L2", 256, 135, 240), ("L3", 512, 68, 120)]:
        f1 = F.normalize(torch.randn(2, C, H, W, generator=g), dim=1).to(dev)
        f2 = F.normalize(torch.randn(2, C, H, W, generator=g), dim=1).to(dev)
        nb = 4 * 2 * H * W * (2 * C + 81)
        for name, fn in (("raw", lambda: correlation.forward(f1, f2, 1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)),
                         ("fused", lambda: correlation.local_correlation_layer(f2, f1))):
            us = timeit(fn)
            print(f"  {lvl} {name:6s} {us:8.1f} us  {nb / us / 1e3:8.1f} GB/s  {nb / us / 1e3 / 8000:.3f} of 8 TB/s")


if __name__ == "__main__":
    main()
