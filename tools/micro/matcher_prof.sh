export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/matcher; mkdir -p $O
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_m -o m --output-format csv -- python $R/tools/matcher_bench.py --precision fp16 --steps 10 > $O/log.txt 2>&1
cd $R
cp $(find /tmp/prof_m -name "*kernel_stats.csv") $O/kernel_stats.csv
