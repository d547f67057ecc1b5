#!/bin/bash
# tools/quick_check.sh OUT [notests] -- the short form of final_check.sh: GPU test suite, one bench line, rocprofv3 kernel stats of the
# timed region (run on the GPU box; ~15 min with the tests, ~6 without)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-quick}; mkdir -p $O
if [ "$2" != "notests" ]; then timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.txt; fi
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_n1.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --no-cpu --steps 6 --warmup 3 > /tmp/prof_bench.log 2>&1
cd $R
grep '^{"metric"' /tmp/prof_bench.log > $O/bench_under_rocprof.json
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_bench_kernel_stats.csv \;
python tools/trace_window_stats.py $(find /tmp/prof_bench -name "*kernel_trace.csv") $O/bench_under_rocprof.json $O/rocprofv3_bench_kernel_stats_timed_region.csv > $O/trace_window.txt
cat $O/pytest_gpu.txt $O/bench_n1.json 2>/dev/null | cut -c1-1500
