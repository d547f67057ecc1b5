#!/usr/bin/env python3
"""tools/step_async.py -- is the Refign step launch-bound?  Per phase: host time spent ENQUEUEING (no synchronisation
inside the step) next to the whole step's wall time; if the enqueue total is close to the step time, the GPU is waiting
for the host."""
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
ap.add_argument("--steps", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
if "RANK" in os.environ:                                   # same path as bench.py under torchrun (1 rank here)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
wl = bench.WORKLOADS[a.workload](dev, 2, 1234, a.height, a.width, a.precision)
m = wl.model
acc = {}


def timed(name, fn):
    t0 = time.perf_counter()
    out = fn()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return out


for attr in ("align", "refine", "get_dacs_mix", "update_momentum_encoder", "calc_feat_dist"):
    orig = getattr(m, attr)
    setattr(m, attr, (lambda o, n: (lambda *x, **k: timed(n, lambda: o(*x, **k))))(orig, attr))
for name in ("m_backbone", "m_head", "backbone", "head"):
    mod = getattr(m, name)
    orig = mod.forward
    mod.forward = (lambda o, n: (lambda *x, **k: timed(n + ".fwd", lambda: o(*x, **k))))(orig, name)
bw = m.manual_backward
m.manual_backward = lambda loss, retain_graph=False: timed("backward", lambda: bw(loss, retain_graph))
ost = m._optimizer.step
m._optimizer.step = lambda: timed("allreduce+adamw", ost)

for i in range(a.steps):
    acc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"step {i}: enqueue {1e3 * (t1 - t0):7.1f} ms  total {1e3 * (t2 - t0):7.1f} ms   host-side per phase: "
          + "  ".join(f"{k}={v:.1f}" for k, v in acc.items()), flush=True)

if os.environ.get("RFN_SYNC_DEBUG", "0") == "1":          # list the host<->device synchronisation points of one step
    import warnings
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    wl.step()
    torch.cuda.set_sync_debug_mode("default")

if os.environ.get("RFN_SYNC_DEBUG", "0") == "2":          # which autograd node synchronises inside backward()?
    state = {"node": None}

    def trace(loss):
        seen, stack = set(), [loss.grad_fn]
        while stack:
            n = stack.pop()
            if n is None or n in seen:
                continue
            seen.add(n)
            n.register_prehook(lambda g, n=n: state.__setitem__("node", n.name()))
            stack.extend(f for f, _ in n.next_functions)
        return len(seen)

    def bw_debug(loss, retain_graph=False):
        nn_ = trace(loss)
        torch.cuda.set_sync_debug_mode("error")
        try:
            bw(loss, retain_graph)
            print(f"backward over {nn_} nodes: no synchronising op", flush=True)
        except RuntimeError as e:
            print(f"backward over {nn_} nodes: SYNC inside node {state['node']}: {str(e)[:200]}", flush=True)
            raise SystemExit(0)
        finally:
            torch.cuda.set_sync_debug_mode("default")

    m.manual_backward = bw_debug
    wl.step()
