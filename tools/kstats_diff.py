#!/usr/bin/env python3
"""tools/kstats_diff.py old.csv new.csv [steps_old steps_new] -- per-kernel time per step of two rocprofv3 --stats summaries
(timed-region CSVs of tools/trace_window_stats.py), largest differences first."""
import csv
import re
import sys


def load(path, steps):
    out = {}
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Name"])[:110]
        t = out.setdefault(name, [0.0, 0.0])
        t[0] += float(r["TotalDurationNs"]) / 1e6 / steps
        t[1] += int(r["Calls"]) / steps
    return out


so, sn = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (5.0, 5.0)
a, b = load(sys.argv[1], so), load(sys.argv[2], sn)
print(f"total: {sum(v[0] for v in a.values()):.2f} ms / {sum(v[1] for v in a.values()):.0f} launches  ->  "
      f"{sum(v[0] for v in b.values()):.2f} ms / {sum(v[1] for v in b.values()):.0f} launches per step")
rows = sorted(((b.get(k, [0, 0])[0] - a.get(k, [0, 0])[0], k) for k in set(a) | set(b)), key=lambda r: -abs(r[0]))
for d, k in rows[:int(sys.argv[5]) if len(sys.argv) > 5 else 30]:
    x, y = a.get(k, [0, 0]), b.get(k, [0, 0])
    print(f"{d:+8.3f} ms  {x[0]:8.3f} ({x[1]:6.0f}) -> {y[0]:8.3f} ({y[1]:6.0f})  {k}")
