#!/bin/bash
# round 5, fifth call: whole GPU suite (short tracebacks), stream priority of the step's main work A/B
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05e; mkdir -p $O
RFN_TEST_REPORT_DIR=$O timeout 1800 python -m pytest tests -x -q -m gpu --tb=short 2>&1 | tail -40 > $O/pytest_gpu.txt
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['ms_per_step'])"; }
{
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for i in 1 2; do
echo -n "default stream (default)           : "; run
echo -n "RFN_MAIN_PRIORITY=0 (own stream)   : "; RFN_MAIN_PRIORITY=0 bash -c "$(declare -f run); run"
echo -n "RFN_MAIN_PRIORITY=1 (low)          : "; RFN_MAIN_PRIORITY=1 bash -c "$(declare -f run); run"
done
} > $O/main_priority_ab.txt 2>&1
for f in pytest_gpu.txt align_amp_1080x1920.txt main_priority_ab.txt; do echo "== $f"; cat $O/$f; done
