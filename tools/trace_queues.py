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

# ---- timeline of the last step: how busy is each queue in 5 ms buckets, and where does the busiest queue idle ----
ts = marks[-2], marks[-1]
last = [r for r in rows if ts[0] <= r[0] < ts[1]]
qs = sorted(byq, key=lambda q: -sum(e - s for s, e, _, _ in byq[q]))[:2]
B = 5e6
nb = int((ts[1] - ts[0]) / B) + 1
print(f"\nlast step ({(ts[1] - ts[0]) / 1e6:.1f} ms): busy fraction per 5 ms bucket, queues {qs}")
for q in qs:
    occ = [0.0] * nb
    for s, e, _, qq in last:
        if qq != q:
            continue
        b0 = int((s - ts[0]) / B)
        while s < e and b0 < nb:
            be = ts[0] + (b0 + 1) * B
            occ[b0] += min(e, be) - s
            s = min(e, be)
            b0 += 1
    print(f"  {str(q):12s} " + " ".join(f"{int(round(100 * o / B)):3d}" for o in occ))
main = sorted([r for r in last if r[3] == qs[0]])
gaps = []
for a, b in zip(main, main[1:]):
    if b[0] - a[1] > 150e3:
        gaps.append((b[0] - a[1], a, b))
print(f"idle gaps > 150 us on {qs[0]}: {len(gaps)}, total {sum(g[0] for g in gaps) / 1e6:.1f} ms")
for g, a, b in sorted(gaps, key=lambda x: -x[0])[:12]:
    print(f"  {g / 1e3:8.0f} us at +{(a[1] - ts[0]) / 1e6:6.1f} ms   after {a[2][:60]}   before {b[2][:60]}")
print("\ndominant kernel per 10 ms bucket of the last step:")
for q in qs:
    print(f" queue {q}")
    nb2 = int((ts[1] - ts[0]) / 1e7) + 1
    for b in range(nb2):
        agg2 = defaultdict(int)
        for s, e, k, qq in last:
            if qq == q and int((s - ts[0]) / 1e7) == b:
                agg2[k] += e - s
        if agg2:
            top = sorted(agg2.items(), key=lambda kv: -kv[1])[:2]
            print(f"   +{b * 10:4d} ms  busy {sum(agg2.values()) / 1e6:5.1f}  " + " | ".join(f"{t / 1e6:4.1f} {k[:70]}" for k, t in top))
