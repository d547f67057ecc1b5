#!/bin/bash
# tools/corr_kt.sh [target] -- rocprofv3 kernel-trace duration of the correlation forward selected by RFN_CORR_VARIANT /
# RFN_CORR_MFMA_CFG (pure kernel time; HIP events around back-to-back Python launches include host gaps)
export TMPDIR=/tmp
R=$PWD
T=${1:-corr_l1_fused}
cd /tmp
d=/tmp/kt_${T}_$$
timeout 300 rocprofv3 --kernel-trace --stats -d $d -o kt --output-format csv -- python $R/tools/prof_corr.py $T 12 > /tmp/kt_$T.log 2>&1
python - "$(find $d -name '*kernel_stats.csv' | head -1)" corr9 <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print(f"kernel-trace: {r['Name'][:100]}  calls={r['Calls']}  avg={float(r['AverageNs']) / 1e3:.1f} us  min={float(r['MinNs']) / 1e3:.1f} us  max={float(r['MaxNs']) / 1e3:.1f} us")
PY
