#!/usr/bin/env python3
"""tools/align_ab.py -- align() + refine() at 1080x1920, b=2 under the step's autocast (fp16 matcher convolutions):
time per call and the top kernels.  RFN_CONV_MFMA=0 puts the dense convolutions back on the ROCm library (A/B)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.RefignStep(dev, 2, 1234, 1080, 1920, "bf16")
m, b = wl.model, wl.batch
logits = torch.randn(2, 19, 1080, 1920, device=dev)


def run():
    w, mask, cert = m.align(logits, b["image_ref"], b["image_trg"])
    return m.refine(logits, w, mask, cert)


with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("align+refine ms:", e0.elapsed_time(e1) / 5, " RFN_CONV_MFMA =", os.environ.get("RFN_CONV_MFMA", "1"))
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run()
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=22, max_name_column_width=90))
