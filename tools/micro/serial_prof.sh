export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/serial; mkdir -p $O
cat > /tmp/serial_bench.py <<'PY'
import sys, runpy
sys.argv = ['bench.py', '--no-cpu', '--no-roofline', '--steps', '4', '--warmup', '4']
sys.path.insert(0, "/root/repo")
from refign_amd import uda
uda.DomainAdaptationSegmentationModel._overlap_teacher = lambda self, x: False
runpy.run_path('/root/repo/bench.py', run_name='__main__')
PY
cd $R
timeout 600 python /tmp/serial_bench.py 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial ms/step', d['ms_per_step'])" > $O/serial.txt
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_serial -o s --output-format csv -- python /tmp/serial_bench.py > /tmp/prof_serial.log 2>&1
cd $R
grep '^{"metric"' /tmp/prof_serial.log > $O/bench_under_rocprof.json
python tools/trace_window_stats.py $(find /tmp/prof_serial -name "*kernel_trace.csv") $O/bench_under_rocprof.json $O/kernel_stats_timed_region.csv > $O/trace_window.txt
