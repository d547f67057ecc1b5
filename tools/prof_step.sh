#!/bin/bash
# tools/prof_step.sh NAME [bench args...] -- rocprofv3 kernel trace of `bench.py --no-cpu --steps 6 --warmup 3`, reduced to the timed
# region (tools/trace_window_stats.py): gpurun_out/NAME_timed_region.csv, NAME_bench.json, NAME_window.txt.  Raw trace stays in /tmp.
name=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$name
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o bench --output-format csv -- python $R/bench.py --no-cpu --no-roofline --steps 6 --warmup 3 "$@" > /tmp/prof_$name.log 2>&1
grep '^{"metric"' /tmp/prof_$name.log > $R/gpurun_out/${name}_bench.json
python $R/tools/trace_window_stats.py $(find /tmp/prof_$name -name "*kernel_trace.csv") $R/gpurun_out/${name}_bench.json $R/gpurun_out/${name}_timed_region.csv | tee $R/gpurun_out/${name}_window.txt
