#!/bin/bash
# tools/micro/ab_r04b.sh -- in-step A/B of the 192 x 256 second-generation GEMM tiles and the tile-count threshold (one box)
run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
python -m pytest tests/test_mfma_gpu.py -x -q -k "second_generation" 2>&1 | tail -2
echo -n "RFN_GEMM2=1 (320 only)         : "; RFN_GEMM2=1 run
echo -n "default (320 + 256)            : "; run
echo -n "RFN_GEMM2=1                    : "; RFN_GEMM2=1 run
echo -n "default                        : "; run
echo -n "RFN_GEMM2_MIN_TILES=100        : "; RFN_GEMM2_MIN_TILES=100 run
echo -n "RFN_GEMM2_MIN_TILES=40         : "; RFN_GEMM2_MIN_TILES=40 run
echo -n "RFN_GEMM2_MIN_TILES=400        : "; RFN_GEMM2_MIN_TILES=400 run
echo -n "default again                  : "; run
