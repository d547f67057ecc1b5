#!/usr/bin/env python3
"""tools/step_timeline.py -- where one bench step's phases sit in time (HIP events on the streams the phases run on, no
profiler attached): source forward / backward (main stream), teacher branch (side stream), mixed forward / loss + backward (mix
stream), optimiser + EMA (main stream), and the host time of the step.  Times are ms from the start of the step's EMA update."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.RefignStep(dev, 2, 1234, adapt_to_ref="--adapt-to-ref" in sys.argv)
for _ in range(14 if "--adapt-to-ref" in sys.argv else 8):
    wl.step()
torch.cuda.synchronize()
m = wl.model
ev = {}


def mark(key):
    ev[key] = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))


ALONE = "--alone" in sys.argv          # every phase with the device to itself (synchronize on both sides): its stand-alone time
alone = {}


def wrap(obj, name, key):
    f = getattr(obj, name)

    def g(*a, **k):
        if ALONE:
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = f(*a, **k)
            torch.cuda.synchronize()
            alone.setdefault(key, []).append((time.perf_counter() - t) * 1e3)
            return out
        mark(key + "0")
        out = f(*a, **k)
        mark(key + "1")
        return out
    setattr(obj, name, g)


for gname, key in (("source_pass", "S"), ("mixed_pass", "M")):
    g = m._graphs[gname]
    wrap(g, "forward", key + "f")
    wrap(g, "backward", key + "b")
if "student_backbone" in m._graphs:
    wrap(m._graphs["student_backbone"], "forward", "Bf")
    wrap(m._graphs["student_backbone"], "backward", "Bb")
wrap(m, "_target_branch", "T")
wrap(m, "update_momentum_encoder", "ema")
wrap(m._optimizer, "step", "opt")
if hasattr(m, "prefetch_align_flow"):
    wrap(m, "prefetch_align_flow", "pfA")
    wrap(m, "prefetch_imnet_features", "pfI")
rows = []
for it in range(10 if "--adapt-to-ref" in sys.argv else 5):
    ev.clear()
    t0 = time.perf_counter()
    wl.step()
    host = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    if ALONE:
        names = (("ema", "ema"), ("bb fwd", "Bf"), ("bb bwd", "Bb"), ("src fwd", "Sf"), ("teacher", "T"), ("mix fwd", "Mf"), ("mix bwd", "Mb"), ("src bwd", "Sb"),
                 ("next imnet", "pfI"), ("next flow", "pfA"), ("opt", "opt"))
        print(f"step {it} (phases one at a time): wall {host:6.1f} ms | " +
              " | ".join(f"{lbl} {sum(alone.pop(k)):6.2f}" for lbl, k in names if k in alone))
        continue
    z = ev["ema0"]
    rel = lambda k: z.elapsed_time(ev[k])  # noqa: E731
    parts = [f"{lbl} {rel(k + '0'):6.1f}-{rel(k + '1'):6.1f}" for lbl, k in
             (("ema", "ema"), ("bb fwd", "Bf"), ("src fwd", "Sf"), ("src bwd", "Sb"), ("teacher", "T"), ("mix fwd", "Mf"), ("mix bwd", "Mb"),
              ("bb bwd", "Bb"), ("opt", "opt"))
             if k + "0" in ev]
    print(f"step {it}: host {host:5.1f} ms | " + " | ".join(parts))
print("(a phase's start is when its first kernel may run on its stream, its end when its last kernel has finished; the prefetches of "
      "the next batch follow the teacher branch on the side stream)")
