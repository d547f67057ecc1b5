#!/usr/bin/env python3
"""tools/tail_debug.py -- device timeline of the serial tail of a step (end of the mixed pass -> start of the next source pass):
events recorded on the main stream around every call that enqueues work there."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.RefignStep(dev, 2, 1234)
for _ in range(6):
    wl.step()
torch.cuda.synchronize()
m = wl.model
marks = []


def mark(name):
    marks.append((name, torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True))))


def wrap(obj, name, key):
    f = getattr(obj, name)

    def g(*a, **k):
        mark(key + ":in")
        out = f(*a, **k)
        mark(key + ":out")
        return out
    setattr(obj, name, g)


class G:
    def __init__(self, g, key):
        self.g, self.key = g, key

    def __call__(self, *a):
        mark(self.key + ":in")
        out = self.g(*a)
        mark(self.key + ":out")
        return out

    def __getattr__(self, k):
        return getattr(self.g, k)


m._graphs["source_pass"] = G(m._graphs["source_pass"], "S")
m._graphs["mixed_pass"] = G(m._graphs["mixed_pass"], "M")
t = m._optimizer.t if hasattr(m._optimizer, "t") else None
wrap(t.grads, "merge_second", "merge")
wrap(t.grads, "zero", "zero")
wrap(t.fast_step, "step", "adamw")
wrap(m, "update_momentum_encoder", "ema")
wrap(m, "_take_class_prefetch", "classes_take")
wrap(m, "_prefetch_classes", "classes_next")
wrap(m, "prefetch_imnet_features", "imnet_prefetch")
for it in range(3):
    marks.clear()
    wl.step()
    wl.step()
    torch.cuda.synchronize()
    i0 = max(i for i, (n, _) in enumerate(marks) if n == "M:out" and any(k == "S:in" for k, _ in marks[i:]))
    base = marks[i0][1]
    print(f"-- step boundary {it}")
    for n, e in marks[i0:]:
        print(f"   {n:22s} {base.elapsed_time(e):8.3f} ms")
        if n == "S:in":
            break
