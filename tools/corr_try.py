#!/usr/bin/env python3
"""tools/corr_try.py [reps] -- the level-1 fused local correlation (C = 128, 270 x 480, b = 2) timed with HIP events, spaced
launches (as bench.py's roofline leg) and back to back, for the kernel configuration selected by the RFN_CORR_* variables
of the environment (they are read once per process: one process per configuration, tools/corr_try.sh loops)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refign_amd import correlation  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
b, C, H, W = 2, 128, 270, 480
# the step's operands are L2-normalised ReLU features (VGG): about half the entries are zero
f1 = torch.nn.functional.normalize(torch.relu(torch.randn(b, C, H, W, generator=g)), dim=1).to(dev)
f2 = torch.nn.functional.normalize(torch.relu(torch.randn(b, C, H, W, generator=g)), dim=1).to(dev)
fn = lambda: correlation.local_correlation_layer(f2, f1)  # noqa: E731
ref = fn()
for _ in range(3):
    fn()


def series(spaced):
    ev = []
    torch.cuda.synchronize()
    for _ in range(reps):
        if spaced:
            torch.cuda._sleep(2_000_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b_) * 1e3 for a, b_ in ev)
    return sum(t) / len(t), t[len(t) // 2], t[0]


sp, b2b = series(True), series(False)
byts = 4 * b * H * W * (2 * C + 81)
cfg = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RFN_CORR"))
print(f"{cfg or 'default':40s} spaced avg {sp[0]:7.1f} med {sp[1]:7.1f} min {sp[2]:7.1f} us ({byts / sp[0] / 1e6 / 8:.3f} of 8 TB/s)   "
      f"back-to-back avg {b2b[0]:7.1f} min {b2b[2]:7.1f}   checksum {float(ref.double().sum()):.6f}")
