#!/usr/bin/env python3
"""tools/corr_small_maps.py -- the fused local correlation layer on the tiny maps of the pyramid (K4 level 3, K2 levels 2 / 3): the
cross-workgroup channel split (one launch, joined by the tile's last workgroup) against the one-kernel path (RFN_CORR_SPLIT=0);
HIP events around each call, spaced (bench.launch_series_us)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from refign_amd import correlation
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (C, H, W) in [(256, 32, 32), (256, 64, 64), (512, 32, 32)]:
    f1 = torch.nn.functional.normalize(torch.randn(2, C, H, W, generator=g), dim=1).to(dev)
    f2 = torch.nn.functional.normalize(torch.randn(2, C, H, W, generator=g), dim=1).to(dev)
    for split in ("1", "0"):
        os.environ["RFN_CORR_SPLIT"] = split
        fn = lambda: correlation.local_correlation_layer(f2, f1)
        for _ in range(3):
            fn()
        print(f"2x{C}x{H}x{W} RFN_CORR_SPLIT={split} (splits {correlation._channel_splits(2, C, H, W)}): {bench.launch_series_us(fn, True):.1f} us per call (HIP events, spaced)", flush=True)
