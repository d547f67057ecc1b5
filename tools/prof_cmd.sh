#!/bin/bash
# tools/prof_cmd.sh <name> <runs> <rows> -- <command ...>: rocprofv3 kernel trace + stats of a command (output in /tmp on the GPU
# box: the raw database is hundreds of MB), the per-kernel summary copied to gpurun_out/<name>_kernel_stats.csv and its top rows printed.
name=$1; runs=$2; rows=$3; shift 4
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p --output-format csv -- "$@" > /tmp/prof_$name.log 2>&1
echo "rc $?"; grep -v "rocprofv3\|^W2026\|^E2026" /tmp/prof_$name.log | tail -5
f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
cp $f $root/gpurun_out/${name}_kernel_stats.csv && python $root/tools/kstats.py $f $runs $rows
