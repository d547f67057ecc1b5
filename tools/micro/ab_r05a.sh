#!/bin/bash
# round 5, first contact: step tests for the split passes / merged backward / early mixed forward, then the in-call A/B of the bench
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_dacs_gpu.py -x -q -m gpu > gpurun_out/r05a_tests.txt 2>&1
tail -5 gpurun_out/r05a_tests.txt
for cfg in "1 1" "0 0" "1 0" "1 1" "0 0"; do
  set -- $cfg
  RFN_MERGE_FD_BACKWARD=$1 RFN_EARLY_MIXED_FWD=$2 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2> gpurun_out/r05a_err_$1$2.txt | tail -1 | python -c "
import sys,json
l=json.loads(sys.stdin.read()); print('merge=$1 early=$2', l['ms_per_step'], l['value'], l['config'].get('hipgraph_regions'))" >> gpurun_out/r05a_ab.txt
done
cat gpurun_out/r05a_ab.txt
