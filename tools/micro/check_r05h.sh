#!/bin/bash
# round 5, eighth call: K/V-resident attention forward -- tests, kernel bench at several thresholds, step A/B
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05h; mkdir -p $O
timeout 1500 python -m pytest tests/test_mfma_gpu.py tests/test_race_gpu.py tests/test_seg_gpu.py -x -q -m gpu --tb=short 2>&1 | tail -12 > $O/pytest_subset.txt
{ echo "# resident kernel from 8192 query blocks on (default)"; timeout 300 python tools/attn_bench.py 2>&1 | grep -v "amdgpu.ids";
  echo "# resident kernel everywhere it fits (RFN_ATTN_RES_MIN=0)"; RFN_ATTN_RES_MIN=0 timeout 300 python tools/attn_bench.py 2>&1 | grep -v "amdgpu.ids";
  echo "# streaming kernel only (RFN_ATTN_RES_MIN=1000000000)"; RFN_ATTN_RES_MIN=1000000000 timeout 300 python tools/attn_bench.py 2>&1 | grep -v "amdgpu.ids"; } > $O/attn_bench_resident.txt
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['ms_per_step'])"; }
{
for i in 1 2; do
echo -n "K/V-resident attention forward (default: >= 8192 blocks) : "; run
echo -n "   everywhere it fits (RFN_ATTN_RES_MIN=0)                : "; RFN_ATTN_RES_MIN=0 bash -c "$(declare -f run); run"
echo -n "   from 2000 blocks on                                    : "; RFN_ATTN_RES_MIN=2000 bash -c "$(declare -f run); run"
echo -n "streaming kernel only                                     : "; RFN_ATTN_RES_MIN=1000000000 bash -c "$(declare -f run); run"
done
} > $O/attn_resident_ab.txt 2>&1
for f in pytest_subset.txt attn_bench_resident.txt attn_resident_ab.txt; do echo "== $f"; cat $O/$f; done
