#!/bin/bash
# tools/micro/host_bound.sh -- is a loop host-bound?  Total kernel time under rocprofv3 --kernel-trace against the loop's wall time, for the
# evaluation forward (N2) and the matcher training step (N1).  (kernel time / wall well below 1 with one stream = the device waits for the host)
export TMPDIR=/tmp
R=$PWD
cd /tmp
for t in "eval_bench.py --steps 10" "matcher_bench.py --precision fp16 --steps 10 --warmup 3"; do
  n=$(echo $t | cut -d. -f1)
  rm -rf /tmp/hb_$n
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/hb_$n -o hb --output-format csv -- python $R/tools/$t > /tmp/hb_$n.log 2>&1
  grep "ms/batch\|ms/step" /tmp/hb_$n.log
  python - "$(find /tmp/hb_$n -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"   all launches of the process (warm-up included): {calls} kernels, {tot / 1e6:.1f} ms of kernel time, {tot / calls / 1e3:.1f} us per kernel")
for r in rows[:6]:
    print(f"   {float(r['TotalDurationNs']) / 1e6:8.1f} ms {int(r['Calls']):7d}  {r['Name'][:100]}")
PY
done
