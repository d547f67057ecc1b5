#!/usr/bin/env python3
"""tools/trace_window_stats.py TRACE.csv BENCH.json OUT.csv [MARKER] -- per-kernel statistics (rocprofv3 --stats
columns + CallsPerStep) of the STEADY-STATE part of a `rocprofv3 --kernel-trace -- python bench.py` run, so that
warm-up (MIOpen's first-call solver search with its naive reference convolutions, JIT) and the set-up of the roofline
micro-benchmark do not drown the steps being timed.  The window is cut with a marker kernel that runs exactly once per
step (default: rfn::align_tail_kernel): from its first launch in the timed steps to its last one = (steps - 1) whole
step periods.  rocprofv3's own whole-process statistics are kept next to it."""
import csv
import json
import sys
from collections import defaultdict

trace, bench, out = sys.argv[1:4]
marker = sys.argv[4] if len(sys.argv) > 4 else "align_tail_kernel"
line = json.loads(open(bench).read().strip().splitlines()[-1])
steps = int(line["steps"])
rows = []
with open(trace) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
marks = sorted(s for s, _, k in rows if marker in k)
assert len(marks) >= steps >= 2, (len(marks), steps)
t0, t1 = marks[-steps], marks[-1]
periods = steps - 1
agg = defaultdict(list)
for s, e, k in rows:
    if t0 <= s < t1:
        agg[k].append(e - s)
tot = sum(sum(v) for v in agg.values())
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "CallsPerStep"])
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), sum(v), sum(v) / len(v), round(100.0 * sum(v) / tot, 4), min(v), max(v),
                    round(len(v) / periods, 2)])
print(f"{periods} step periods, {(t1 - t0) / 1e6 / periods:.1f} ms/step under the tracer, "
      f"{sum(len(v) for v in agg.values()) / periods:.0f} launches/step, kernel time {tot / 1e6 / periods:.1f} ms/step "
      f"({100 * tot / (t1 - t0):.1f}% of the period)")
