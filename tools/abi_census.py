#!/usr/bin/env python3
"""tools/abi_census.py -- every call of the C ABI in one eager Refign step, with its integer arguments and its device
time: which entry point at which shape the step spends its kernel time in.

The ctypes library object is replaced by a proxy (here, in the tool -- the product path is untouched) that brackets each
call with two events on the stream the call launches on.  The step runs without graphs and without the stream overlap
(one stream: an event pair then measures the kernels of that call and nothing else); the gap between two events around
an empty region (~2-4 us here) is measured first and subtracted.

  python tools/abi_census.py [--top 60] [--entry rfn_gemm_nt]     ->  profiles/rNN_abi_census.txt by redirect
"""
import argparse
import collections
import os
import sys

os.environ.setdefault("RFN_HIP_GRAPH", "0")
os.environ.setdefault("RFN_MIXED_CONCURRENT", "0")

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from refign_amd import _lib  # noqa: E402



def _short(a):
    if isinstance(a, float):
        return a
    if a is None:
        return 0
    if isinstance(a, int):
        return a if abs(a) < (1 << 31) else "p"
    return "p"                                                    # ctypes arrays / byte strings


class Proxy:
    def __init__(self, lib):
        self._lib = lib
        self.on = False
        self.records = []

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("rfn_") or name in ("rfn_last_error", "rfn_abi_version"):
            return fn

        def call(*args):
            if not self.on:
                return fn(*args)
            ints = tuple(_short(a) for a in args[:-1])            # last argument: the stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            self.records.append((name, ints, e0, e1))
            return rc
        return call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--entry", default=None, help="list every shape of this entry point")
    ap.add_argument("--precision", default="bf16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS["refign_hrda_step_1080x1920"](dev, 2, 1234, 1080, 1920, a.precision)
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    proxy = Proxy(_lib.load_library())
    _lib._lib = proxy
    # the empty event pair
    gaps = []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        gaps.append((e0, e1))
    torch.cuda.synchronize()
    gap = sorted(x.elapsed_time(y) for x, y in gaps)[100] * 1e3
    proxy.on = True
    wl.step()
    torch.cuda.synchronize()
    proxy.on = False
    agg = collections.OrderedDict()
    for name, ints, e0, e1 in proxy.records:
        us = max(e0.elapsed_time(e1) * 1e3 - gap, 0.0)
        k = (name, ints)
        c = agg.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += us
    per_entry = collections.Counter()
    calls = collections.Counter()
    for (name, ints), (n, us) in agg.items():
        per_entry[name] += us
        calls[name] += n
    total = sum(per_entry.values())
    print(f"# one eager step, one stream: {len(proxy.records)} ABI calls, {total / 1e3:.1f} ms inside them "
          f"(empty event pair {gap:.1f} us, subtracted)")
    print("# entry point                              calls     ms")
    for name, us in per_entry.most_common():
        print(f"{name:42s} {calls[name]:6d} {us / 1e3:8.2f}")
    print(f"\n# top {a.top} (entry, integer arguments in ABI order; p = pointer, 0 = null)      calls   us each     ms")
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    for (name, ints), (n, us) in rows[:a.top]:
        print(f"{name:30s} {str(ints):110s} {n:5d} {us / n:9.1f} {us / 1e3:8.2f}")
    if a.entry:
        print(f"\n# every shape of {a.entry}")
        for (name, ints), (n, us) in rows:
            if name == a.entry:
                print(f"{str(ints):120s} {n:5d} {us / n:9.1f} {us / 1e3:8.2f}")


if __name__ == "__main__":
    main()
