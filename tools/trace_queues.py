#!/usr/bin/env python3
"""tools/trace_queues.py TRACE.csv BENCH.json [MARKER] -- steady-state window of a rocprofv3 kernel trace of bench.py,
per hardware queue: busy time (union of kernel intervals), kernel time and launches per step, the top kernels of each
queue, plus a family breakdown (hand-written rfn:: / hipBLASLt / MIOpen+CK / attention libraries / ATen elementwise)."""
import csv
import json
import sys
from collections import defaultdict

trace, bench = sys.argv[1:3]
marker = sys.argv[3] if len(sys.argv) > 3 else "align_tail_kernel"
steps = int(json.loads(open(bench).read().strip().splitlines()[-1])["steps"])
rows = []
with open(trace) as f:
    rd = csv.DictReader(f)
    cols = rd.fieldnames
    qcol = "Queue_Id" if "Queue_Id" in cols else None
    scol = "Stream_Id" if "Stream_Id" in cols else None
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     (r.get(qcol, "?") if qcol else "?", r.get(scol, "?") if scol else "?")))
print("columns:", cols)
marks = sorted(s for s, _, k, _ in rows if marker in k)
t0, t1 = marks[-steps], marks[-1]
periods = steps - 1
win = [r for r in rows if t0 <= r[0] < t1]


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs if cs is not None else 0)


def family(k):
    if "rfn::" in k:
        return "hand-written rfn::"
    if k.startswith("Cijk_"):
        return "hipBLASLt"
    if "attn_fwd" in k or "aiter" in k or "fmha" in k:
        return "attention libraries"
    if "igemm" in k or "miopen" in k.lower() or "ck::" in k or "_ZN2ck" in k or "batched_transpose" in k or "SubTensor" in k \
            or "MIOpen" in k:
        return "MIOpen / CK"
    if "at::native" in k or "at_cuda_detail" in k:
        return "ATen"
    return "other"


print(f"{periods} periods, {(t1 - t0) / 1e6 / periods:.1f} ms/step, all-queue busy (union) "
      f"{union([(s, e) for s, e, _, _ in win]) / 1e6 / periods:.1f} ms/step")
fam = defaultdict(lambda: [0, 0])
for s, e, k, _ in win:
    fam[family(k)][0] += e - s
    fam[family(k)][1] += 1
for k, (t, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"  family {k:22s} {t / 1e6 / periods:8.2f} ms/step  {n / periods:8.0f} launches/step")
byq = defaultdict(list)
for r in win:
    byq[r[3]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _, _ in kv[1])):
    kt = sum(e - s for s, e, _, _ in rs)
    print(f"queue/stream {q}: kernel time {kt / 1e6 / periods:.1f} ms/step, busy {union([(s, e) for s, e, _, _ in rs]) / 1e6 / periods:.1f} "
          f"ms/step, {len(rs) / periods:.0f} launches/step")
    agg = defaultdict(lambda: [0, 0])
    for s, e, k, _ in rs:
        agg[k][0] += e - s
        agg[k][1] += 1
    for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:22]:
        print(f"      {t / 1e6 / periods:8.2f} ms  n={n / periods:7.1f}  avg {t / n / 1e3:8.1f} us  {k[:110]}")
