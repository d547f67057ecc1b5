#!/bin/bash
# tools/micro/ab_r04c.sh -- in-step A/B of the balanced persistent grid of the second-generation GEMM (one box, alternating)
run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
python -m pytest tests/test_mfma_gpu.py -x -q -k "second_generation" 2>&1 | tail -1
for i in 1 2 3; do
echo -n "RFN_GEMM2_BALANCE=0 (256 workgroups) : "; RFN_GEMM2_BALANCE=0 run
echo -n "balanced grid (default)             : "; run
done
