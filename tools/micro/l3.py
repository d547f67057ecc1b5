import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from refign_amd import correlation
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (C, H, W) in [(256, 32, 32), (512, 17, 30), (256, 64, 64), (512, 32, 32)]:
    f1 = torch.nn.functional.normalize(torch.randn(2, C, H, W, generator=g), dim=1).to(dev)
    f2 = torch.nn.functional.normalize(torch.randn(2, C, H, W, generator=g), dim=1).to(dev)
    for split in ("1", "0"):
        os.environ["RFN_CORR_SPLIT"] = split
        fn = lambda: correlation.local_correlation_layer(f2, f1)
        for _ in range(3):
            fn()
        print(f"2x{C}x{H}x{W} RFN_CORR_SPLIT={split} (splits {correlation._channel_splits(2, C, H, W)}): {bench.launch_series_us(fn, True):.1f} us per call (HIP events, spaced)", flush=True)
