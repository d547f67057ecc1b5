#!/usr/bin/env python3
"""tools/micro/corr_race.py [reps [b C H W]] -- repeat the level-1 fused local correlation (2 x 128 x 270 x 480) and compare every result with
the first, bit for bit; RFN_CORR_* select the kernel (read once per process).  A counted-wait pipeline that reads a ring slot
before its DMA has landed shows up here as a handful of differing pixels in a few launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from refign_amd import correlation  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
shape = tuple(int(v) for v in sys.argv[2:6]) if len(sys.argv) >= 6 else (2, 128, 270, 480)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
b, C, H, W = shape
f1 = torch.nn.functional.normalize(torch.relu(torch.randn(b, C, H, W, generator=g)), dim=1).to(dev)
f2 = torch.nn.functional.normalize(torch.relu(torch.randn(b, C, H, W, generator=g)), dim=1).to(dev)
torch.cuda.synchronize()
ref = correlation.local_correlation_layer(f2, f1).clone()
torch.cuda.synchronize()
bad = 0
worst = 0.0
for i in range(reps):
    if i % 3 == 0:   # vary what else the device is doing: a big copy in flight on another stream
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            junk = f1.clone()
    out = correlation.local_correlation_layer(f2, f1)
    if not torch.equal(out, ref):
        bad += 1
        d = (out - ref).abs()
        worst = max(worst, float(d.max()))
        if bad <= 3:
            idx = torch.nonzero(d > 0)
            print(f"  launch {i}: {idx.shape[0]} differing values, max {float(d.max()):.3e}, first at {idx[0].tolist()}")
cfg = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RFN_CORR"))
print(f"{cfg or 'default':40s} {b}x{C}x{H}x{W} {reps} launches: {bad} differ from the first (max |diff| {worst:.3e}); checksum {float(ref.double().sum()):.6f}")
