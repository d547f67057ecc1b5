#!/usr/bin/env python3
"""tools/overlap_debug.py -- when do the source pass (main stream) and the teacher branch (side stream) of a step start
and end on the device, and what is the host doing meanwhile?  CUDA events around both + host clock."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
if "RANK" in os.environ:                                   # same path as bench.py under torchrun (1 rank here)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
wl = bench.RefignStep(dev, 2, 1234)
for _ in range(6):
    wl.step()
torch.cuda.synchronize()
m = wl.model
ev = {}
host = {}
orig_graph = m._graphs["source_pass"]
orig_branch = m._target_branch


class Wrap:
    def __call__(self, *a):
        host["S_call"] = time.perf_counter()
        ev["S0"] = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
        out = orig_graph(*a)
        ev["S1"] = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
        host["S_ret"] = time.perf_counter()
        return out

    def __getattr__(self, k):
        return getattr(orig_graph, k)


def branch(batch):
    host["T_call"] = time.perf_counter()
    ev["T0"] = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
    out = orig_branch(batch)
    ev["T1"] = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
    host["T_ret"] = time.perf_counter()
    return out


m._graphs["source_pass"] = Wrap()
m._target_branch = branch
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0 = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
    wl.step()
    e1 = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
    host["step_ret"] = time.perf_counter()
    torch.cuda.synchronize()
    rel = lambda k: e0.elapsed_time(ev[k])
    print(f"step {it}: device total {e0.elapsed_time(e1):.1f} ms | S {rel('S0'):6.1f} -> {rel('S1'):6.1f} | T {rel('T0'):6.1f} -> {rel('T1'):6.1f} ms")
    print("         host: " + "  ".join(f"{k} {1e3 * (v - t0):6.1f}" for k, v in sorted(host.items(), key=lambda kv: kv[1])))
