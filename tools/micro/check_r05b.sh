#!/bin/bash
# round 5, second call: new tests (K2 golden, RFN_DDP_MODE rehearsal), N=1 with and without a 1-rank group, K2 line, the
# timed-region kernel statistics of the restructured step
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_syncbn_gpu.py tests/test_align_gpu.py tests/test_dacs_gpu.py tests/test_step_gpu.py -x -q -m gpu 2>&1 | tail -4 > $O/pytest_subset.txt
timeout 600 python bench.py --no-cpu 2>$O/bench_n1.err | tail -1 > $O/bench_n1.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-cpu --no-roofline 2>$O/bench_torchrun_n1.err | grep '^{"metric"' > $O/bench_torchrun_n1.json
timeout 600 python bench.py --no-cpu --no-roofline 2>/dev/null | tail -1 > $O/bench_n1_again.json
timeout 600 python bench.py --workload uawarpc_align_512x512 --steps 50 --warmup 5 2>$O/bench_k2.err | tail -1 > $O/bench_k2.json
timeout 600 python bench.py --gpus 2 --no-cpu > $O/bench_gpus2_refused.txt 2>&1; echo "rc=$?" >> $O/bench_gpus2_refused.txt
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --no-cpu --steps 6 --warmup 3 > /tmp/prof_bench.log 2>&1
cd $R
grep '^{"metric"' /tmp/prof_bench.log > $O/bench_under_rocprof.json
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_bench_kernel_stats.csv \;
python tools/trace_window_stats.py $(find /tmp/prof_bench -name "*kernel_trace.csv") $O/bench_under_rocprof.json $O/rocprofv3_bench_kernel_stats_timed_region.csv > $O/trace_window.txt
python tools/trace_queues.py $(find /tmp/prof_bench -name "*kernel_trace.csv") $O/bench_under_rocprof.json > $O/queues.txt 2>&1
timeout 300 python tools/opt_phase_debug.py 2>&1 | grep "^step" > $O/step_phases_events.txt
for f in pytest_subset.txt bench_n1.json bench_torchrun_n1.json bench_n1_again.json bench_k2.json bench_gpus2_refused.txt trace_window.txt; do echo "== $f"; cut -c1-600 $O/$f; done
