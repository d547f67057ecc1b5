#!/bin/bash
# tools/micro/ab_r04i.sh OUT -- the end-of-round defaults (RFN_BN_WGS=1024, RFN_GEMM_NT_MIN_TILES=2000, RFN_FUSED_GELU_BWD=1) against
# the previous ones, alternating; then the whole GPU suite and the bench line as the driver runs it
cd "$(dirname "$0")/../.."
O=gpurun_out/${1:-r04i}; mkdir -p $O
run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
for i in 1 2 3; do
  for cfg in "" "RFN_BN_WGS=2048 RFN_GEMM_NT_MIN_TILES=1000 RFN_FUSED_GELU_BWD=0"; do
    echo -n "${cfg:-default (1024 / 2000 / 1)} : "; env $cfg bash -c "$(declare -f run); run"
  done
done > $O/knob_ab.txt 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_n1.json
timeout 600 python bench.py --precision k5 --no-cpu 2>/dev/null | tail -1 > $O/bench_k5_n1.json
timeout 600 python bench.py --workload refign_daformer_step_1080x1920 --height 512 --width 1024 --no-cpu --no-roofline 2>/dev/null | tail -1 > $O/bench_k3_daformer_512x1024.json
