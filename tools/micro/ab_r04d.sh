#!/bin/bash
# tools/micro/ab_r04d.sh -- in-step sweep: how many CUs the teacher's persistent GEMMs should leave to the other streams
run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
for c in 256 240 224 208 192 256 240 224; do echo -n "RFN_GEMM2_GRID_CAP=$c : "; RFN_GEMM2_GRID_CAP=$c run; done
