#!/bin/bash
# tools/corr_ablate.sh VARIANT -- level-1 correlation launch time with parts of the kernel switched off
# (RFN_CORR_ABLATE bits: 1 no DMA, 2 no arithmetic, 4 no stores; results are meaningless then)
for a in 0 1 2 4 3 5 6 7; do
  echo "== variant $1 ablate $a"
  RFN_CORR_VARIANT=$1 RFN_CORR_ABLATE=$a timeout 120 python tools/corr_variant_check.py 2>&1 | grep "L1\|L2"
done
