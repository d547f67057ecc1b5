#!/usr/bin/env python3
"""tools/micro/stream_probe.py -- do the streams of a step run CONCURRENTLY?  After a few bench steps (so that every
stream / graph / communicator exists), a spin kernel is put on each pair of streams; a pair that shares an in-order
hardware queue takes twice the time of one kernel.  Run plain and with RANK=0 WORLD_SIZE=1 (1-rank RCCL group)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
if "RANK" in os.environ:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
wl = bench.RefignStep(dev, 2, 1234)
for _ in range(6):
    wl.step()
torch.cuda.synchronize()
m = wl.model
streams = {"main": torch.cuda.current_stream(), "side": getattr(m, "_side_stream", None),
           "mix": getattr(m, "_mix_stream", None)}
for i in range(4):
    streams[f"pool{i}"] = torch.cuda.Stream()
streams = {k: v for k, v in streams.items() if v is not None}
CY = int(3e7)


def span(ss):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    ends = []
    e0.record(ss[0])
    for s in ss[1:]:
        s.wait_event(e0)
    for s in ss:
        with torch.cuda.stream(s):
            torch.cuda._sleep(CY)
            e = torch.cuda.Event(enable_timing=True)
            e.record(s)
            ends.append(e)
    torch.cuda.synchronize()
    return max(e0.elapsed_time(e) for e in ends)


one = span([streams["main"]])
print(f"one spin kernel: {one:.2f} ms")
names = list(streams)
for i, a in enumerate(names):
    for b in names[i + 1:]:
        t = span([streams[a], streams[b]])
        print(f"  {a:6s} + {b:6s}: {t:6.2f} ms  {'CONCURRENT' if t < 1.5 * one else 'serial'}")
print(f"  main+side+mix: {span([streams[k] for k in ('main', 'side', 'mix') if k in streams]):.2f} ms")
