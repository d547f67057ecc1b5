export TMPDIR=/tmp
R=$PWD
for cfg in "0 0" "30 0" "30 2"; do
  set -- $cfg
  d=/tmp/prof_ab_$1_$2
  (cd /tmp && RFN_CORR_VARIANT=$1 RFN_CORR_MFMA_CFG=$2 timeout 600 rocprofv3 --kernel-trace --stats -d $d -o bench --output-format csv -- python $R/bench.py --no-cpu --steps 6 --warmup 3 > /tmp/ab_$1_$2.log 2>&1)
  echo "== variant $1 cfg $2"
  grep -o '"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"avg_launch_us": [0-9.]*' /tmp/ab_$1_$2.log | tr '\n' ' '; echo
  python - "$(find $d -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "corr9" in r["Name"]:
        print(f"  {r['Name'][:80]}  calls={r['Calls']}  avg={float(r['AverageNs']) / 1e3:.1f} us  min={float(r['MinNs']) / 1e3:.1f}  max={float(r['MaxNs']) / 1e3:.1f}")
PY
done
