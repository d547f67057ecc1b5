#!/bin/bash
# tools/micro/ab_r04g.sh OUT -- end-of-round check of HEAD (GPU suite, the bench line as the driver runs it, rocprofv3 kernel stats of
# the timed region), then the in-step launch-geometry knobs re-swept once more against the round's final kernels (one run each,
# the default configuration at the start, in the middle and at the end: the spread of those three is the noise floor)
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
R=$PWD
O=$R/gpurun_out/${1:-r04g}; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/pytest_gpu.txt
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_n1.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --no-cpu --steps 6 --warmup 3 > /tmp/prof_bench.log 2>&1
cd $R
grep '^{"metric"' /tmp/prof_bench.log > $O/bench_under_rocprof.json
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_bench_kernel_stats.csv \;
python tools/trace_window_stats.py $(find /tmp/prof_bench -name "*kernel_trace.csv") $O/bench_under_rocprof.json $O/rocprofv3_bench_kernel_stats_timed_region.csv > $O/trace_window.txt
run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
{
  for cfg in "" "RFN_ATTN_DKV_WGS=96" "RFN_ATTN_DKV_WGS=192" "RFN_GEMM_TN_MIN_WGS=384" "RFN_GEMM_TN_MIN_WGS=768" \
             "RFN_DWCONV_SLICED_BLOCKS=48" "RFN_DWCONV_SLICED_BLOCKS=96" "" "RFN_GEMM2_MIN_TILES=120" "RFN_GEMM2_MIN_TILES=400" \
             "RFN_GEMM_NT_MIN_TILES=600" "RFN_GEMM_NT_MIN_TILES=2000" "RFN_FUSED_GELU_BWD=1" "RFN_BN_WGS=1024" "RFN_SIDE_PRIORITY=0" ""; do
    echo -n "${cfg:-default} : "; env $cfg bash -c "$(declare -f run); run"
  done
} > $O/knob_sweep.txt 2>&1
