#!/bin/bash
# tools/refresh_profiles.sh [all|rocprof] -- regenerate the measurement artefacts that get copied into profiles/ (run on the GPU box)
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
if [ "${1:-all}" = "all" ]; then
  timeout 600 python tools/kbench.py 2>&1 | grep -v "amdgpu.ids\|MIOpen" > gpurun_out/kbench_final.log
  timeout 500 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_n1.json
  timeout 400 python tools/step_profile.py --rows 70 2>&1 | grep -v "amdgpu.ids\|MIOpen" > gpurun_out/step_profile.txt
  timeout 300 python tools/step_phases.py --steps 3 2>&1 | grep "^step\|max mem" > gpurun_out/step_phases.txt
fi
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --no-cpu --steps 6 --warmup 3 > /tmp/prof_bench.log 2>&1
cd $R
grep '^{"metric"' /tmp/prof_bench.log > gpurun_out/bench_under_rocprof.json
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/rocprofv3_bench_kernel_stats.csv \;
python tools/trace_window_stats.py $(find /tmp/prof_bench -name "*kernel_trace.csv") gpurun_out/bench_under_rocprof.json gpurun_out/rocprofv3_bench_kernel_stats_timed_region.csv
ls -la gpurun_out /tmp/prof_bench | head -30
