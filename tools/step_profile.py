#!/usr/bin/env python3
"""tools/step_profile.py -- torch.profiler kernel table of one Refign step (after warm-up)."""
import argparse
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--workload", default="refign_hrda_step_1080x1920")
ap.add_argument("--rows", type=int, default=45)
a = ap.parse_args()
dev = torch.device("cuda:0")
wl = bench.WORKLOADS[a.workload](dev, 2, 1234, a.height, a.width, a.precision)
for _ in range(2):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    wl.step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=a.rows, max_name_column_width=90))
