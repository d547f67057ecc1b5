#!/usr/bin/env python3
"""tools/corr_trace.py -- per-workgroup phase timeline of the level-1 correlation launch (RFN_CORR_TRACE hook of
corr_mfma.hip): when do workgroups start, how long do they wait for LDS-DMA, when do they store."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

path = "/tmp/corr_trace.bin"
os.environ["RFN_CORR_TRACE"] = path
os.environ.setdefault("RFN_CORR_VARIANT", "30")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refign_amd import correlation  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
C, H, W = 128, 270, 480
f1 = F.normalize(torch.randn(2, C, H, W, generator=g), dim=1).to(dev)
f2 = F.normalize(torch.randn(2, C, H, W, generator=g), dim=1).to(dev)
for fused in (False, True):
    for _ in range(int(os.environ.get("RFN_CORR_TRACE_EVERY", "1")) * 3):
        (correlation.local_correlation_layer(f2, f1) if fused else correlation.forward(f1, f2, 1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1))
    torch.cuda.synchronize()
    r = np.fromfile(path, dtype=np.int64).reshape(-1, 8).astype(np.float64)
    t0 = r[:, 0].min()
    us = lambda x: x / 100.0   # 100 MHz
    start, wait, loop, alu, end = us(r[:, 0] - t0), us(r[:, 1]), us(r[:, 2] - t0), us(r[:, 3] - t0), us(r[:, 4] - t0)
    q = lambda a: "min %7.1f  p10 %7.1f  med %7.1f  p90 %7.1f  max %7.1f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
    print(f"fused={fused}  {len(r)} workgroups, span {end.max():.1f} us")
    print("  start            ", q(start))
    print("  loop end         ", q(loop))
    print("  loop duration    ", q(loop - start))
    print("  of which waiting ", q(wait))
    print("  epilogue ALU+issue", q(alu - loop))
    print("  store drain      ", q(end - alu))
    print("  end              ", q(end))
    print("  shader clock over the loop (s_memtime ticks / 100 MHz ticks): %.0f MHz" % np.median(r[:, 7] / (r[:, 2] - r[:, 0]) * 100.0))
    xcc = r[:, 5].astype(int) & 0xf
    for x in range(8):
        m = xcc == x
        if m.any():
            print(f"  xcc {x}: {m.sum():3d} wgs  loop {np.median((loop - start)[m]):6.1f}  wait {np.median(wait[m]):6.1f}  end {np.median(end[m]):6.1f}")
