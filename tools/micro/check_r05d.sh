#!/bin/bash
# round 5, fourth call: grouped weight gradients (tests + in-call A/B), the timed precision map at 1080x1920, step timeline
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05d; mkdir -p $O
RFN_TEST_REPORT_DIR=$O timeout 1500 python -m pytest tests/test_mfma_gpu.py tests/test_step_gpu.py tests/test_align_gpu.py tests/test_seg_gpu.py tests/test_params_gpu.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_subset.txt
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['ms_per_step'])"; }
{
for i in 1 2 3; do
echo -n "grouped weight gradients (default) : "; run
echo -n "RFN_GROUP_WGRADS=0                 : "; RFN_GROUP_WGRADS=0 bash -c "$(declare -f run); run"
done
} > $O/group_wgrads_ab.txt 2>&1
timeout 300 python tools/step_timeline.py 2>&1 | grep "^step\|^(" > $O/step_timeline.txt
for f in pytest_subset.txt align_amp_1080x1920.txt group_wgrads_ab.txt step_timeline.txt; do echo "== $f"; cat $O/$f; done
