#!/bin/bash
# tools/micro/corr_p2.sh -- the software-pipelined 4-stage correlation kernel (default at level 1; RFN_CORR_VARIANT=40 forces it)
# against the first 4-stage kernel (20) and the 2-stage one (16): launch time, ablations, and the repeat-and-compare race check
cd "$(dirname "$0")/../.."
for cfg in "RFN_CORR_VARIANT=20" "" "RFN_CORR_VARIANT=20" "" "RFN_CORR_XCD=0" "RFN_CORR_XCD=0" "RFN_CORR_ABLATE=1" "RFN_CORR_ABLATE=4" "RFN_CORR_ABLATE=5" "$@"; do
  env $cfg python tools/corr_try.py 30
done
for cfg in "" "RFN_CORR_XCD=0" "RFN_CORR_VARIANT=20" "RFN_CORR_VARIANT=16"; do
  env $cfg python tools/micro/corr_race.py ${RACE_REPS:-1000}
done
