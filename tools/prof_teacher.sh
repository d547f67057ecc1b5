#!/bin/bash
# tools/prof_teacher.sh NAME -- rocprofv3 kernel statistics of the gradient-free half alone (bench.py --workload
# refign_align_refine_1080x1920: EMA teacher on 40 views, align, refine), per step: gpurun_out/NAME_kernel_stats.csv
name=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$name
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o bench --output-format csv -- python $R/bench.py --workload refign_align_refine_1080x1920 --no-cpu --no-roofline --steps 6 --warmup 3 "$@" > /tmp/prof_$name.log 2>&1
grep '^{"metric"' /tmp/prof_$name.log > $R/gpurun_out/${name}_bench.json
python $R/tools/trace_window_stats.py $(find /tmp/prof_$name -name "*kernel_trace.csv") $R/gpurun_out/${name}_bench.json $R/gpurun_out/${name}_timed_region.csv | tee $R/gpurun_out/${name}_window.txt
python $R/tools/kstats.py $R/gpurun_out/${name}_timed_region.csv 5 40 | cut -c1-200
