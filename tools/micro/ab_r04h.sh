#!/bin/bash
# tools/micro/ab_r04h.sh -- the three knobs that came out of ab_r04g.sh's single runs below the noise floor's lower edge
# (RFN_BN_WGS=1024, RFN_GEMM_NT_MIN_TILES=2000, RFN_FUSED_GELU_BWD=1), each and together, alternating with the default, 3 runs each
cd "$(dirname "$0")/../.."
run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
for i in 1 2 3; do
  for cfg in "" "RFN_BN_WGS=1024" "RFN_GEMM_NT_MIN_TILES=2000" "RFN_FUSED_GELU_BWD=1" "RFN_BN_WGS=1024 RFN_GEMM_NT_MIN_TILES=2000 RFN_FUSED_GELU_BWD=1"; do
    echo -n "${cfg:-default} : "; env $cfg bash -c "$(declare -f run); run"
  done
done
