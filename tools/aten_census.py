#!/usr/bin/env python3
"""tools/aten_census.py -- which ATen operators (i.e. NOT the C ABI) still launch kernels in one eager Refign step: operator,
input shapes, calls, device time.  torch.profiler on one step without graphs and without the stream overlap."""
import os
import sys

os.environ.setdefault("RFN_HIP_GRAPH", "0")
os.environ.setdefault("RFN_MIXED_CONCURRENT", "0")

import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
# aten_census.py [workload [height width]] -- default: the bench line's HRDA step; K3: refign_daformer_step_1080x1920 512 1024
name = sys.argv[1] if len(sys.argv) > 1 else "refign_hrda_step_1080x1920"
hh, ww = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1080, 1920)
os.environ.setdefault("RFN_GRAPH_STUDENT", "0")
wl = bench.WORKLOADS[name](dev, 2, 1234, hh, ww, "bf16")
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    wl.step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0 and e.key.startswith("aten::"):
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"# ATen operators with device time in one eager step: {tot / 1e3:.1f} ms in {sum(r[1] for r in rows)} calls")
print("#     ms  calls  operator                          input shapes")
for dt, n, k, sh in rows[:int(os.environ.get("TOP", "70"))]:
    print(f"{dt / 1e3:8.2f} {n:6d}  {k:32s}  {sh}")
