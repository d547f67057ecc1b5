#!/bin/bash
# tools/micro/corr_l2.sh -- the smaller correlation levels (kbench rows L2 / L3 / K2-L1): default (8 x 32 single tiles on the
# software-pipelined kernel, 4-stage ring) against the 8-stage variant (RFN_CORR_VARIANT=46) and the 2-stage
# 8 x 64 kernel (3), and the repeat-and-compare race check
cd "$(dirname "$0")/../.."
for v in 0 46 0 46 3; do echo "== RFN_CORR_VARIANT=$v"; RFN_CORR_VARIANT=$v python tools/kbench.py --only L2,L3,K2-L1 2>&1 | grep "corr9 +relu+l2norm\|corr9 raw" | grep -v "L1 "; done
python tools/micro/corr_race.py 300 2 256 135 240
python tools/micro/corr_race.py 300 2 128 128 128
python tools/micro/corr_race.py 300 1 64 71 52
