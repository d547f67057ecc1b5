export TMPDIR=/tmp
R=$PWD
timeout 600 python tools/matcher_bench.py $MB_ARGS 2>&1 | grep -v "MIOpen\|amdgpu.ids" | tail -3
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_m -o m --output-format csv -- python $R/tools/matcher_bench.py --steps 5 --warmup 2 $MB_ARGS > /tmp/m.log 2>&1
python - "$(find /tmp/prof_m -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 7e6:.1f} ms/step (7 steps), {sum(int(r['Calls']) for r in rows) / 7:.0f} launches/step")
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs']) / 7e6:8.2f} ms/step  n={int(r['Calls']) / 7:7.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:110]}")
PY
