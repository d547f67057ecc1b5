#!/bin/bash
# tools/final_check.sh OUT -- what gets copied into profiles/ at the end of a round (run on the GPU box): the GPU test suite, the
# bench lines (headline, parity mode, K5, K2, K3, torchrun N = 1), the step timeline from HIP events, rocprofv3 kernel statistics of
# the timed region, the correlation kernel's counter passes, kernel micro-benchmarks.  PMC passes never share a run with tracing.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-final}; mkdir -p $O
RFN_TEST_REPORT_DIR=$O timeout 1800 python -m pytest tests -x -q -m gpu --tb=short 2>&1 | tail -40 > $O/pytest_gpu.txt
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_n1.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-cpu --no-roofline 2>/dev/null | grep '^{"metric"' > $O/bench_torchrun_n1.json
timeout 900 python bench.py --precision fp32 --no-cpu --steps 10 2>/dev/null | tail -1 > $O/bench_fp32_n1.json
timeout 600 python bench.py --precision k5 --no-cpu 2>/dev/null | tail -1 > $O/bench_k5_n1.json
timeout 600 python bench.py --adapt-to-ref --no-cpu 2>/dev/null | tail -1 > $O/bench_adapt_to_ref_n1.json
timeout 600 python bench.py --workload uawarpc_align_512x512 --steps 50 --warmup 5 2>/dev/null | tail -1 > $O/bench_k2.json
timeout 600 python bench.py --workload refign_daformer_step_1080x1920 --height 512 --width 1024 --no-cpu --no-roofline 2>/dev/null | tail -1 > $O/bench_k3_daformer_512x1024.json
timeout 300 python tools/step_timeline.py 2>&1 | grep "^step\|^(" > $O/step_timeline.txt
timeout 300 python tools/step_timeline.py --adapt-to-ref 2>&1 | grep "^step\|^(" > $O/step_timeline_adapt_to_ref.txt
{ timeout 300 python tools/step_timeline.py --alone 2>&1 | grep "^step"; timeout 300 python tools/step_timeline.py --alone --adapt-to-ref 2>&1 | grep "^step"; } > $O/phases_alone_final.txt
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --no-cpu --steps 6 --warmup 3 > /tmp/prof_bench.log 2>&1
cd $R
grep '^{"metric"' /tmp/prof_bench.log > $O/bench_under_rocprof.json
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_bench_kernel_stats.csv \;
python tools/trace_window_stats.py $(find /tmp/prof_bench -name "*kernel_trace.csv") $O/bench_under_rocprof.json $O/rocprofv3_bench_kernel_stats_timed_region.csv > $O/trace_window.txt
python tools/trace_queues.py $(find /tmp/prof_bench -name "*kernel_trace.csv") $O/bench_under_rocprof.json > $O/queues.txt 2>&1
bash tools/pmc_corr.sh corr_l1_fused > $O/pmc_corr9.txt 2>&1
python tools/pmc_corr_json.py $O/pmc_corr9.txt $O/pmc_traffic_corr9.json > /dev/null 2>&1
timeout 300 python tools/kbench.py --only L1,L2,L3,K2-L1 2>&1 | grep -v "amdgpu.ids\|MIOpen" > $O/kbench_corr.txt
timeout 300 python tools/kbench.py --only tail 2>&1 | grep -v "amdgpu.ids\|MIOpen" >> $O/kbench_corr.txt
timeout 300 python tools/corr_small_maps.py 2>&1 | grep -v "amdgpu.ids" > $O/corr_small_maps.txt
timeout 300 python tools/attn_bench.py 2>&1 | grep -v "amdgpu.ids" > $O/attn_bench.txt
timeout 300 python tools/ffn_bench.py 2>&1 | grep stage > $O/ffn_bench_final.txt
timeout 400 python tools/align_precision_layers.py "timed map" "fp32 everything" 2>&1 | grep -v "amdgpu.ids" > $O/align_precision_timed_map.txt
bash tools/prof_teacher.sh final_teacher_half > /dev/null 2>&1; cp $R/gpurun_out/final_teacher_half_timed_region.csv $O/teacher_half_kernel_stats_timed_region.csv
bash tools/ddp_rehearsal.sh > $O/ddp_rehearsal_final.txt 2>&1
for t in gemm_fc1_s3:gemm_nt gemm_fc2_s3:gemm_nt conv_bottleneck:gemm_nt attn_fwd_s3:attn_fwd wgrad_s3:gemm_tn ffn_s3:ffn_fc1; do
  echo "== ${t%%:*}"; bash tools/pmc_mfma.sh ${t%%:*} ${t##*:}; done > $O/pmc_mfma_kernels.txt 2>&1
{ timeout 300 python tools/matcher_bench.py --precision fp16 2>&1 | tail -2; } > $O/matcher_bench.txt
timeout 300 python tools/aten_census.py 2>&1 | grep -v "amdgpu.ids\|Warn\|_warn" > $O/aten_census.txt
