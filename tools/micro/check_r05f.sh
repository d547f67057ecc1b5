#!/bin/bash
# round 5, sixth call: where this step's matcher flow / the next batch's image-only work run -- A/B
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05f; mkdir -p $O
timeout 900 python -m pytest tests/test_step_gpu.py -x -q -m gpu --tb=short 2>&1 | tail -5 > $O/pytest_step.txt
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['ms_per_step'])"; }
{
for i in 1 2; do
echo -n "default (flow prefetched a step ahead, side stream) : "; run
echo -n "RFN_ALIGN_FLOW_ON=mix                               : "; RFN_ALIGN_FLOW_ON=mix bash -c "$(declare -f run); run"
echo -n "RFN_PREFETCH_ON=main                                : "; RFN_PREFETCH_ON=main bash -c "$(declare -f run); run"
echo -n "RFN_ALIGN_FLOW_ON=mix RFN_PREFETCH_ON=main          : "; RFN_ALIGN_FLOW_ON=mix RFN_PREFETCH_ON=main bash -c "$(declare -f run); run"
done
} > $O/flow_placement_ab.txt 2>&1
RFN_ALIGN_FLOW_ON=mix timeout 300 python tools/step_timeline.py 2>&1 | grep "^step" > $O/step_timeline_flow_on_mix.txt
for f in pytest_step.txt flow_placement_ab.txt step_timeline_flow_on_mix.txt; do echo "== $f"; cat $O/$f; done
