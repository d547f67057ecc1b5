#!/bin/bash
# round 5, tenth call: pruned knobs + ADVICE fixes + BatchNorm eligibility -- whole GPU suite, matcher bench, bench line
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05j; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu --tb=short 2>&1 | tail -25 > $O/pytest_gpu.txt
{ timeout 300 python tools/matcher_bench.py --precision fp16 2>&1 | tail -2; } > $O/matcher_bench.txt
timeout 600 python bench.py --no-cpu 2>/dev/null | tail -1 > $O/bench_n1.json
for f in pytest_gpu.txt matcher_bench.txt bench_n1.json; do echo "== $f"; cut -c1-700 $O/$f; done
