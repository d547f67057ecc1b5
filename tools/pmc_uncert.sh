#!/bin/bash
# PMC passes for the fused uncertainty kernel (separate passes, no tracing domains combined with --pmc)
export TMPDIR=/tmp
R=$PWD
cd /tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "FETCH_SIZE WRITE_SIZE"; do
  d=/tmp/pmc_unc_$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $c -d $d -o pmc --output-format csv -- python $R/tools/prof_corr.py uncert_l1 4 > /dev/null 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'uncert9' in r.get('Kernel_Name', '')]
acc = collections.defaultdict(list)
for r in rows:
    acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f"{k:32s} mean_per_dispatch={sum(v) / len(v):.6g}  n={len(v)}  kernel=rfn::uncert9_frontend_kernel (L1: 2 x 81 x 270 x 480)")
PY
done
