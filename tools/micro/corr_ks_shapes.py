"""tools/micro/corr_ks_shapes.py -- the fused local-correlation layer on maps of 34 ... 272 tiles of 8 x 32 (C = 256), timed with HIP
events over back-to-back launches: how does a launch scale with the number of workgroups when every CU holds at most one?
Run under RFN_CORR_VARIANT=41 (one 3-wave workgroup per tile) and =47 (two 3-wave groups per tile, half the channels each)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from refign_amd import correlation  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (B, C, H, W) in [(1, 256, 32, 240), (1, 256, 64, 240), (1, 256, 135, 240), (2, 256, 96, 240), (2, 256, 128, 240), (2, 256, 135, 240),
                     (2, 128, 135, 240), (2, 64, 135, 240)]:
    f1 = torch.nn.functional.normalize(torch.relu(torch.randn(B, C, H, W, generator=g)), dim=1).to(dev)
    f2 = torch.nn.functional.normalize(torch.relu(torch.randn(B, C, H, W, generator=g)), dim=1).to(dev)
    for _ in range(10):
        correlation.local_correlation_layer(f2, f1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        correlation.local_correlation_layer(f2, f1)
    e1.record()
    torch.cuda.synchronize()
    tiles = B * ((H + 7) // 8) * ((W + 31) // 32)
    print(f"variant {os.environ.get('RFN_CORR_VARIANT', '0'):>2}  {B} x {C} x {H} x {W}: {tiles:4d} tiles  {e0.elapsed_time(e1) / n * 1e3:7.1f} us")
