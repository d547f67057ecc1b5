#!/bin/bash
# round 5, third call: correlation (paired edge tiles at level 2, pruned library), new gcorr_post, low-priority prefetch stream A/B
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05c; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_race_gpu.py tests/test_align_gpu.py tests/test_abi_cpu.py -x -q 2>&1 | tail -6 > $O/pytest_subset.txt
timeout 300 python tools/kbench.py --only L1,L2,L3,K2-L1 2>&1 | grep -v "amdgpu.ids\|MIOpen" > $O/kbench_corr.txt
timeout 300 python tools/kbench.py --only tail 2>&1 | grep -v "amdgpu.ids\|MIOpen" | grep "global corr" >> $O/kbench_corr.txt
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['ms_per_step'])"; }
{
for i in 1 2; do
echo -n "default                                         : "; run
echo -n "GPU_MAX_HW_QUEUES=8                             : "; GPU_MAX_HW_QUEUES=8 bash -c "$(declare -f run); run"
echo -n "GPU_MAX_HW_QUEUES=8 RFN_PREFETCH_STREAM=1 prio 0 : "; GPU_MAX_HW_QUEUES=8 RFN_PREFETCH_STREAM=1 RFN_PREFETCH_PRIORITY=0 bash -c "$(declare -f run); run"
echo -n "GPU_MAX_HW_QUEUES=8 RFN_PREFETCH_STREAM=1 prio 1 : "; GPU_MAX_HW_QUEUES=8 RFN_PREFETCH_STREAM=1 RFN_PREFETCH_PRIORITY=1 bash -c "$(declare -f run); run"
done
echo -n "RFN_PREFETCH_STREAM=1 prio 0 (4 queues)          : "; RFN_PREFETCH_STREAM=1 RFN_PREFETCH_PRIORITY=0 bash -c "$(declare -f run); run"
} > $O/prefetch_stream_ab.txt 2>&1
for f in pytest_subset.txt kbench_corr.txt prefetch_stream_ab.txt; do echo "== $f"; cat $O/$f; done
