#!/bin/bash
# tools/micro/ab_r04e.sh -- does the second-generation GEMM pay on the student's one-round launches (M = 8160, N = 1280: 172 tiles)?
cd "$(dirname "$0")/../.."
echo "## per shape (replayed graph of 20 launches), us: RFN_GEMM2_MIN_TILES = 200 (default) / 100 / 40"
for t in 200 100 40; do echo -n "min_tiles=$t "; RFN_GEMM2_MIN_TILES=$t SWEEP_CFGS="" SWEEP_PERSIST=0 python tools/gemm_sweep.py | tail -1; done
SWEEP_CFGS="x" SWEEP_PERSIST="" python tools/gemm_sweep.py | head -1
run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
echo "## step"
for t in 200 100 200 100 40; do echo -n "RFN_GEMM2_MIN_TILES=$t : "; RFN_GEMM2_MIN_TILES=$t run; done
