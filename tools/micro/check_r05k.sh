#!/bin/bash
# round 5: mix stream priority with more hardware queues (experiment)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05k; mkdir -p $O
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['ms_per_step'])"; }
{
for i in 1 2; do
echo -n "default                                         : "; run
echo -n "RFN_X_MIX_PRIORITY=-1                           : "; RFN_X_MIX_PRIORITY=-1 bash -c "$(declare -f run); run"
echo -n "RFN_X_MIX_PRIORITY=-1 GPU_MAX_HW_QUEUES=8       : "; GPU_MAX_HW_QUEUES=8 RFN_X_MIX_PRIORITY=-1 bash -c "$(declare -f run); run"
done
} > $O/mix_priority_ab.txt 2>&1
cat $O/mix_priority_ab.txt
