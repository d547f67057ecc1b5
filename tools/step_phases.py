#!/usr/bin/env python3
"""tools/step_phases.py -- wall time of the phases of one Refign step at a given size (synchronised between phases)."""
import argparse
import os
import sys
import time

import torch

os.environ.setdefault("RFN_HIP_GRAPH", "0")      # per-phase host timing needs the eager path

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--workload", default="refign_hrda_step_1080x1920")
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
wl = bench.WORKLOADS[a.workload](dev, 2, 1234, a.height, a.width, a.precision)
m = wl.model


def timed(name, fn, store):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    store[name] = store.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return out


# wrap the phases
acc = {}
for attr in ("align", "refine", "get_dacs_mix", "update_momentum_encoder", "calc_feat_dist"):
    orig = getattr(m, attr)
    setattr(m, attr, (lambda o, n: (lambda *x, **k: timed(n, lambda: o(*x, **k), acc)))(orig, attr))
for name in ("m_backbone", "m_head", "backbone", "head"):
    mod = getattr(m, name)
    orig = mod.forward
    mod.forward = (lambda o, n: (lambda *x, **k: timed(n + ".fwd", lambda: o(*x, **k), acc)))(orig, name)
bw = m.manual_backward
m.manual_backward = lambda loss, retain_graph=False: timed("backward", lambda: bw(loss, retain_graph), acc)
ost = m._optimizer.step
m._optimizer.step = lambda: timed("allreduce+adamw", ost, acc)

for i in range(a.steps):
    acc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.step()
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) * 1e3
    print(f"step {i}: {tot:8.1f} ms   " + "  ".join(f"{k}={v:.1f}" for k, v in acc.items()), flush=True)
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
