#!/bin/bash
# tools/micro/ab_r04k.sh -- further along the two launch-geometry directions that paid at the end of round 4 (smaller GEMM tiles, fewer BatchNorm workgroups)
cd "$(dirname "$0")/../.."
run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
for cfg in "" "RFN_GEMM_NT_MIN_TILES=3000" "RFN_GEMM_NT_MIN_TILES=4000" "RFN_BN_WGS=512" "RFN_BN_STATS_WGS=256" "RFN_GEMM_TN_MIN_WGS=384" "" "RFN_GEMM_NT_MIN_TILES=3000" "RFN_BN_WGS=512" ""; do
  echo -n "${cfg:-default} : "; env $cfg bash -c "$(declare -f run); run"
done
