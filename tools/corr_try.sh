#!/bin/bash
# tools/corr_try.sh -- tools/corr_try.py under a list of kernel configurations (one process each)
cd "$(dirname "$0")/.."
for cfg in "" "RFN_CORR_XCD=1" "RFN_CORR_ABLATE=8" "RFN_CORR_ABLATE=8 RFN_CORR_XCD=1" "RFN_CORR_VARIANT=20" "RFN_CORR_VARIANT=20 RFN_CORR_XCD=1" "RFN_CORR_VARIANT=12" "RFN_CORR_VARIANT=17" "RFN_CORR_VARIANT=5" "RFN_CORR_VARIANT=32" "$@"; do
  env $cfg python tools/corr_try.py 30
done
