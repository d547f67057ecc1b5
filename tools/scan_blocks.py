#!/usr/bin/env python3
"""tools/scan_blocks.py -- time the fused patch-9 correlation at shapes with controlled tile counts (residency scan)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refign_amd import correlation  # noqa: E402

dev = torch.device("cuda:0")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
th = int(os.environ.get("TILE_H", "8"))
for rows_tiles, col_tiles in [(1, 4), (2, 16), (4, 32), (8, 32), (12, 32), (16, 32), (17, 32), (24, 32), (32, 32), (48, 32), (64, 32), (128, 32)]:
    H, W = rows_tiles * th, col_tiles * 64
    a = torch.randn(1, C, H, W, device=dev)
    b = torch.randn(1, C, H, W, device=dev)
    for _ in range(3):
        correlation.local_correlation_layer(b, a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        correlation.local_correlation_layer(b, a)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    nblk = rows_tiles * col_tiles
    fl = 2.0 * 81 * C * H * W
    print(f"blocks={nblk:5d} ({H}x{W})  {us:8.1f} us   {us / nblk * 256:8.1f} us per 256 blocks   {fl / us / 1e6:6.1f} TFLOP/s", flush=True)
