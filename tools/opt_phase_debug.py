#!/usr/bin/env python3
"""tools/opt_phase_debug.py -- device time between the end of the mixed pass and the start of the next source pass
(optimiser step, LR scheduler, EMA update, weight-cache refresh, class set of the next batch), and the host time of
each of those calls."""
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
ev, host = {}, {}


def wrap(obj, name, key):
    f = getattr(obj, name)

    def g(*a, **k):
        t0 = time.perf_counter()
        ev[key + "0"] = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
        out = f(*a, **k)
        ev[key + "1"] = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
        host[key] = host.get(key, 0.0) + (time.perf_counter() - t0) * 1e3
        return out
    setattr(obj, name, g)


class G:
    def __init__(self, g, key):
        self.g, self.key = g, key

    def __call__(self, *a):
        ev[self.key + "0"] = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
        out = self.g(*a)
        ev[self.key + "1"] = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))
        return out

    def __getattr__(self, k):
        return getattr(self.g, k)


m._graphs["source_pass"] = G(m._graphs["source_pass"], "S")
m._graphs["mixed_pass"] = G(m._graphs["mixed_pass"], "M")
wrap(m._optimizer, "step", "opt")
wrap(m, "update_momentum_encoder", "ema")
prev_m1 = None
for it in range(4):
    host.clear()
    t0 = time.perf_counter()
    wl.step()
    host_total = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    msg = f"step {it}: host {host_total:.1f} ms (opt.step {host.get('opt', 0):.1f}, ema {host.get('ema', 0):.1f})"
    msg += f" | device: S {ev['S0'].elapsed_time(ev['S1']):.1f}  M {ev['M0'].elapsed_time(ev['M1']):.1f}  opt {ev['opt0'].elapsed_time(ev['opt1']):.1f}  ema {ev['ema0'].elapsed_time(ev['ema1']):.1f}"
    msg += f"  S-end -> M-start {ev['S1'].elapsed_time(ev['M0']):.1f}"
    if prev_m1 is not None:
        msg += f"  | previous M-end -> this S-start {prev_m1.elapsed_time(ev['S0']):.1f} ms (host-bound here: this tool synchronises every step; tools/tail_debug.py)"
    print(msg)
    prev_m1 = ev["M1"]
