#!/usr/bin/env python3
"""tools/kstats.py <kernel_stats.csv> [runs] [rows] -- the top rows of a rocprofv3 --stats kernel summary, per run."""
import csv
import sys

path, runs, top = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0, int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6 / runs:.3f} ms per run, {sum(int(r['Calls']) for r in rows) / runs:.0f} launches per run")
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / runs:9.3f} ms  n={int(r['Calls']) / runs:7.1f}  avg {float(r['AverageNs']) / 1e3:9.1f} us  "
          f"{r['Name'][:160]}")
