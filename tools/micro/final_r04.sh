#!/bin/bash
# tools/micro/final_r04.sh OUT -- the short end-of-round record at HEAD: GPU suite, smoke, the bench line as the driver runs it, rocprofv3 kernel
# stats of the timed region, K5 / K3 lines, matcher step (the long form is tools/final_check.sh)
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
R=$PWD
O=$R/gpurun_out/${1:-final_r04}; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_n1.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --no-cpu --steps 6 --warmup 3 > /tmp/prof_bench.log 2>&1
cd $R
grep '^{"metric"' /tmp/prof_bench.log > $O/bench_under_rocprof.json
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_bench_kernel_stats.csv \;
python tools/trace_window_stats.py $(find /tmp/prof_bench -name "*kernel_trace.csv") $O/bench_under_rocprof.json $O/rocprofv3_bench_kernel_stats_timed_region.csv > $O/trace_window.txt
python tools/trace_queues.py $(find /tmp/prof_bench -name "*kernel_trace.csv") $O/bench_under_rocprof.json > $O/queues.txt 2>&1
timeout 600 python bench.py --precision k5 --no-cpu 2>/dev/null | tail -1 > $O/bench_k5_n1.json
timeout 600 python bench.py --workload refign_daformer_step_1080x1920 --height 512 --width 1024 --no-cpu --no-roofline 2>/dev/null | tail -1 > $O/bench_k3_daformer_512x1024.json
{ timeout 300 python tools/matcher_bench.py --precision fp16 2>&1 | tail -2; timeout 300 python tools/matcher_bench.py --precision fp32 2>&1 | tail -2; } > $O/matcher_bench.txt
