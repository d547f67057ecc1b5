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
ap.add_argument("--shapes", action="store_true")
ap.add_argument("--cpu", action="store_true", help="rank ops by host (self CPU) time instead of device time")
ap.add_argument("--stacks", default="", help="aten op name: list its call sites (python stacks) by launch count")
ap.add_argument("--pystack", action="store_true", help="--stacks: add the innermost python frames of this repository")
ap.add_argument("--by-time", action="store_true", help="--stacks: order call sites by device time, not by count")
ap.add_argument("--ops", default="", help="comma-separated aten op names for --shapes (default: the dense ops)")
a = ap.parse_args()
dev = torch.device("cuda:0")
if "RANK" in os.environ:                                   # same path as bench.py under torchrun (1 rank here)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
wl = bench.WORKLOADS[a.workload](dev, 2, 1234, a.height, a.width, a.precision)
for _ in range(2):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=a.shapes or bool(a.stacks),
             with_stack=a.pystack) as prof:
    wl.step()
    torch.cuda.synchronize()
if a.stacks:
    # call sites of an op, as the chain of enclosing profiler events (autograd node / aten op / module label)
    from collections import Counter
    cnt, dev_t = Counter(), Counter()
    for e in prof.events():
        if e.name != a.stacks:
            continue
        chain, p = [], e.cpu_parent
        while p is not None and len(chain) < 5:
            chain.append(p.name[:48])
            p = p.cpu_parent
        shp = str(e.input_shapes)[:60] if e.input_shapes else ""
        key = "  <-  ".join(chain) + "   " + shp
        if a.pystack:                                     # innermost python frames inside this repository
            fr = [f for f in (e.stack or []) if "refign_amd/" in f or "bench.py" in f][:2]
            key = " | ".join(f.split("refign_amd/")[-1][:60] for f in fr) + "   " + key
        cnt[key] += 1
        dev_t[key] += e.device_time_total
    order = sorted(cnt, key=lambda k: -dev_t[k]) if a.by_time else [k for k, _ in cnt.most_common()]
    print(f"total: n={sum(cnt.values())} {sum(dev_t.values()) / 1e3:.2f} ms")
    for k in order[:a.rows]:
        print(f"n={cnt[k]:5d} {dev_t[k] / 1e3:8.2f} ms  {k}")
elif a.shapes:
    keys = tuple(a.ops.split(",")) if a.ops else None
    rows = [e for e in prof.key_averages(group_by_input_shape=True)
            if e.key in (keys or ("aten::miopen_convolution", "aten::convolution_backward", "aten::mm", "aten::addmm",
                         "aten::bmm", "aten::copy_", "aten::native_layer_norm", "aten::_flash_attention_forward",
                         "aten::_flash_attention_backward", "aten::native_batch_norm", "aten::upsample_bilinear2d"))]
    rows.sort(key=lambda e: -e.self_device_time_total)
    for e in rows[:a.rows]:
        print(f"{e.self_device_time_total / 1e3:9.2f} ms  n={e.count:4d}  {e.key:34s} {str(e.input_shapes)[:150]}")
elif a.cpu:
    evs = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
    tot = sum(e.self_cpu_time_total for e in evs)
    print(f"total self CPU time {tot / 1e3:.1f} ms (all threads)")
    for e in evs[:a.rows]:
        print(f"{e.self_cpu_time_total / 1e3:9.2f} ms {100 * e.self_cpu_time_total / tot:5.1f}%  n={e.count:5d}  "
              f"{e.self_cpu_time_total / max(e.count, 1):7.1f} us/call  {e.key[:90]}")
else:
    evs = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
    tot = sum(e.self_device_time_total for e in evs)
    print(f"total self device time {tot / 1e3:.1f} ms")
    for e in evs[:a.rows]:
        print(f"{e.self_device_time_total / 1e3:9.2f} ms {100 * e.self_device_time_total / tot:5.1f}%  n={e.count:5d}  {e.key[:110]}")
