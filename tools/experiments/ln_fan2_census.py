import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from refign_amd import seg
dev = torch.device("cuda:0")
blk = seg.Block(320, 5, sr_ratio=2, drop_path=0.0).to(dev).train()
x0 = torch.randn(4, 34 * 60, 320, device=dev)
for flag in (True, False):
    seg._LN_FAN2 = flag
    for it in range(3):
        x = x0.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            xb = x.to(torch.bfloat16)
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
                y = blk(xb, 34, 60)
                y.float().sum().backward()
    ops = {}
    for e in prof.key_averages():
        if e.key.startswith("aten::add") or "Add" in e.key:
            ops[e.key] = e.count
    print("LN_FAN2", flag, ops)
