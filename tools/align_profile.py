#!/usr/bin/env python3
"""tools/align_profile.py -- kernel table of align() + refine() alone at 1080x1920, b=2 (fp32 matcher)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.RefignStep(dev, 2, 1234, 1080, 1920, "bf16")
m = wl.model
b = wl.batch
logits = torch.randn(2, 19, 1080, 1920, device=dev)
with torch.no_grad():
    for _ in range(3):
        w, mask, cert = m.align(logits, b["image_ref"], b["image_trg"])
        p = m.refine(logits, w, mask, cert)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        w, mask, cert = m.align(logits, b["image_ref"], b["image_trg"])
        p = m.refine(logits, w, mask, cert)
    e1.record()
    torch.cuda.synchronize()
    print("align+refine ms:", e0.elapsed_time(e1) / 5)
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
        w, mask, cert = m.align(logits, b["image_ref"], b["image_trg"])
        p = m.refine(logits, w, mask, cert)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=80))
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::miopen_convolution", "aten::conv2d", "aten::cat", "aten::copy_")]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:25]:
    print(f"{e.self_device_time_total / 1e3:9.2f} ms  n={e.count:4d}  {e.key:28s} {str(e.input_shapes)[:120]}")
