export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02j; mkdir -p $O
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python $R/bench.py --no-cpu --no-roofline --steps 6 --warmup 3 > /tmp/prof_bench.log 2>&1
cd $R
grep '^{"metric"' /tmp/prof_bench.log > $O/bench_under_rocprof.json
python tools/trace_window_stats.py $(find /tmp/prof_bench -name "*kernel_trace.csv") $O/bench_under_rocprof.json $O/kernel_stats_timed_region_dist.csv
python tools/trace_queues.py $(find /tmp/prof_bench -name "*kernel_trace.csv") $O/bench_under_rocprof.json > $O/queues_dist.txt 2>&1
