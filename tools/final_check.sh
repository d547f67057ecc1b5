#!/bin/bash
# tools/final_check.sh OUT -- what gets copied into profiles/ at the end of a round: full GPU test suite, bench line,
# step phases from HIP events, rocprofv3 kernel stats of the timed region + per-queue view (run on the GPU box)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-final}; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_n1.json
timeout 300 python tools/opt_phase_debug.py 2>&1 | grep "^step" > $O/step_phases_events.txt
timeout 300 python tools/overlap_debug.py 2>&1 | grep "step\|host" >> $O/step_phases_events.txt
# the serial tail (end of the mixed pass -> start of the next source pass) WITHOUT a host synchronisation between the two steps
# (opt_phase_debug synchronises after every step: its boundary figure includes the host's enqueue time)
timeout 300 python tools/tail_debug.py 2>&1 | grep -v "amdgpu.ids" > $O/step_tail_events.txt
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --no-cpu --steps 6 --warmup 3 > /tmp/prof_bench.log 2>&1
cd $R
grep '^{"metric"' /tmp/prof_bench.log > $O/bench_under_rocprof.json
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_bench_kernel_stats.csv \;
python tools/trace_window_stats.py $(find /tmp/prof_bench -name "*kernel_trace.csv") $O/bench_under_rocprof.json $O/rocprofv3_bench_kernel_stats_timed_region.csv > $O/trace_window.txt
python tools/trace_queues.py $(find /tmp/prof_bench -name "*kernel_trace.csv") $O/bench_under_rocprof.json > $O/queues.txt 2>&1
# round 3: K5 (fp8 teacher) bench line + micro-benchmark + MFMA-pipe counters, correlation kernels (incl. the experimental
# matrix-pipe one and its ablation), DACS kernels
timeout 600 python bench.py --precision k5 --no-cpu 2>/dev/null | tail -1 > $O/bench_k5_n1.json
timeout 300 python tools/f8_bench.py 2>&1 | grep -v "amdgpu.ids" > $O/f8_bench.txt
timeout 300 python tools/kbench.py --only L1,L2 2>&1 | grep -v "amdgpu.ids\|MIOpen" > $O/kbench_corr.txt
for t in f8gemm_fc1_s3:gemm_nt_f8 f8gemm_fc2_s3:gemm_nt_f8 f8attn_s3:attn_fwd_f8; do
  echo "== ${t%%:*}"; bash tools/pmc_mfma.sh ${t%%:*} ${t##*:}; done > $O/pmc_f8_kernels.txt 2>&1
# (kernel-name substrings: "gemm_nt" matches both generations of the NT GEMM -- the teacher's big Linear layers run gemm_nt2_kernel)
for t in gemm_fc1_s3:gemm_nt gemm_fc2_s3:gemm_nt conv_bottleneck:gemm_nt attn_fwd_s3:attn_fwd wgrad_s3:gemm_tn; do
  echo "== ${t%%:*}"; bash tools/pmc_mfma.sh ${t%%:*} ${t##*:}; done > $O/pmc_mfma_kernels.txt 2>&1
# round 3, second half: per-entry-point census of the step, what is left in ATen, attention / depthwise / GEMM sweeps on replayed
# graphs, per-CU fetch rates (L2 / MALL / HBM), cost of a node in a linear graph
timeout 300 python tools/abi_census.py --top 60 2>&1 | grep -v "amdgpu.ids" > $O/abi_census.txt
timeout 300 python tools/aten_census.py 2>&1 | grep -v "amdgpu.ids\|Warn\|_warn" > $O/aten_census.txt
timeout 300 python tools/attn_bench.py 2>&1 | grep -v "amdgpu.ids" > $O/attn_bench.txt
timeout 300 python tools/kbench.py --only dw 2>&1 | grep -v "amdgpu.ids\|MIOpen" > $O/kbench_dwconv.txt
SWEEP_CFGS=";128,128,2;128,64,2;64,64,2;256,256,2" SWEEP_PERSIST=0 timeout 600 python tools/gemm_sweep.py 2>&1 | grep -v "amdgpu.ids" > $O/gemm_sweep.txt
timeout 120 python tools/micro/graph_chain.py 2>&1 | grep -v "amdgpu.ids" > $O/graph_chain.txt
if [ ! -x refign_amd/lib/ab/cu_fetch_rate ]; then mkdir -p refign_amd/lib/ab && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/cu_fetch_rate.hip -o refign_amd/lib/ab/cu_fetch_rate 2>/dev/null; fi
timeout 120 refign_amd/lib/ab/cu_fetch_rate > $O/cu_fetch_rate.txt 2>&1
# round 4: the other BASELINE configurations as bench lines (K3: DAFormer MiT-B5 step at 512x1024, b = 2 -- configs/cityscapes_acdc/
# refign_daformer.yaml:11-41; K5 is written above), the N1 matcher training step, the second-generation GEMM probe, the ASPP and
# correlation experiments, and what one rank of N does (rehearsal + the first-contact script with the one rank there is)
timeout 600 python bench.py --workload refign_daformer_step_1080x1920 --height 512 --width 1024 --no-cpu --no-roofline 2>/dev/null | tail -1 > $O/bench_k3_daformer_512x1024.json
timeout 600 python bench.py --workload refign_daformer_step_1080x1920 --no-cpu --no-roofline 2>/dev/null | tail -1 > $O/bench_daformer_1080x1920.json
{ timeout 300 python tools/matcher_bench.py --precision fp16 2>&1 | tail -2; timeout 300 python tools/matcher_bench.py --precision fp32 2>&1 | tail -2; } > $O/matcher_bench.txt
timeout 600 bash tools/gemm2_probe.sh > $O/gemm2_probe.txt 2>&1
timeout 300 python tools/aspp_try.py 2>&1 | grep -v "amdgpu.ids" > $O/aspp_try_now.txt
timeout 900 bash tools/ddp_rehearsal.sh $(basename $O) > /dev/null 2>&1
timeout 900 bash tools/ddp_matrix.sh $(basename $O) > /dev/null 2>&1
GUARD=240 timeout 1500 bash tools/ddp_first_contact.sh 1 $O/first_contact_1rank > /dev/null 2>&1
bash tools/pmc_corr.sh corr_l1_fused > $O/pmc_corr9.txt 2>&1
python tools/pmc_corr_json.py $O/pmc_corr9.txt $O/pmc_traffic_corr9.json > /dev/null 2>&1
bash tools/micro/corr_p2.sh 2>&1 | grep -v amdgpu.ids > $O/corr_pipe2_now.txt
