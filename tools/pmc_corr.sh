#!/bin/bash
# tools/pmc_corr.sh [prof_corr.py target] -- pipe utilisation, LDS behaviour and HBM traffic of the correlation forward
# one kernel-trace pass + separate PMC passes (never combined with tracing domains).
export TMPDIR=/tmp
R=$PWD
T=${1:-corr_l1_fused}
K=corr9
cd /tmp
d=/tmp/kt_$T
timeout 300 rocprofv3 --kernel-trace --stats -d $d -o kt --output-format csv -- python $R/tools/prof_corr.py $T 6 > /tmp/kt_$T.log 2>&1
python - "$(find $d -name '*kernel_stats.csv' | head -1)" "$K" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print(f"kernel-trace: {r['Name'][:90]}  calls={r['Calls']}  avg={float(r['AverageNs']) / 1e3:.1f} us  min={float(r['MinNs']) / 1e3:.1f} us")
PY
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/pmc_${T}_$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c -d $d -o pmc --output-format csv -- python $R/tools/prof_corr.py $T 4 > /dev/null 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" "$K" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r.get('Kernel_Name', '')]
acc = collections.defaultdict(list)
for r in rows:
    acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f"{k:28s} mean_per_dispatch={sum(v) / len(v):.6g}  n={len(v)}")
PY
done
