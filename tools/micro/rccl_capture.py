#!/usr/bin/env python3
"""tools/micro/rccl_capture.py -- can an RCCL all-reduce be captured into a hipGraph and replayed?  (1-rank group here;
run under torchrun --nproc-per-node N on a multi-GPU node for the real thing)"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
dist.init_process_group("nccl", rank=rank, world_size=world)
x = torch.full((2, 256), float(rank + 1), device="cuda")
w = torch.randn(256, 256, device="cuda")
for _ in range(3):
    y = x @ w
    dist.all_reduce(y)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    y = x @ w
    dist.all_reduce(y)
    z = y * 2
torch.cuda.synchronize()
for i in range(3):
    x.fill_(float(rank + 1 + i))
    g.replay()
    torch.cuda.synchronize()
    want = sum(float(r + 1 + i) for r in range(world)) * w.sum(0) * 2
    print(f"rank {rank} replay {i}: max err {(z[0] - want).abs().max().item():.3e}")
dist.destroy_process_group()
print("capture + replay OK")
