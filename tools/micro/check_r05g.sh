#!/bin/bash
# round 5, seventh call: attention / fp8 kernels with MFMA results in VGPRs (-amdgpu-mfma-vgpr-form) -- tests, kernel bench, step A/B
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05g; mkdir -p $O
timeout 1500 python -m pytest tests/test_mfma_gpu.py tests/test_f8_gpu.py tests/test_race_gpu.py tests/test_seg_gpu.py -x -q -m gpu --tb=short 2>&1 | tail -5 > $O/pytest_subset.txt
timeout 300 python tools/attn_bench.py 2>&1 | grep -v "amdgpu.ids" > $O/attn_bench_vgpr_form.txt
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline $1 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['ms_per_step'])"; }
cp refign_amd/lib/librefign_hip.so /tmp/new.so
{
for i in 1 2; do
cp /tmp/new.so refign_amd/lib/librefign_hip.so
echo -n "MFMA results in VGPRs (attn.hip, f8.hip)  : "; run
echo -n "   RFN_WGRAD_FLUSH_EVERY=4                : "; RFN_WGRAD_FLUSH_EVERY=4 bash -c "$(declare -f run); run"
echo -n "   RFN_WGRAD_FLUSH_EVERY=1000             : "; RFN_WGRAD_FLUSH_EVERY=1000 bash -c "$(declare -f run); run"
cp refign_amd/lib/ab/librefign_hip_noflag.so refign_amd/lib/librefign_hip.so
echo -n "compiler's choice (AGPR accumulators)     : "; run
done
cp /tmp/new.so refign_amd/lib/librefign_hip.so
echo -n "K5, MFMA results in VGPRs                 : "; run "--precision k5"
cp refign_amd/lib/ab/librefign_hip_noflag.so refign_amd/lib/librefign_hip.so
echo -n "K5, compiler's choice                     : "; run "--precision k5"
cp /tmp/new.so refign_amd/lib/librefign_hip.so
} > $O/vgpr_form_ab.txt 2>&1
cp refign_amd/lib/ab/librefign_hip_noflag.so refign_amd/lib/librefign_hip.so
timeout 300 python tools/attn_bench.py 2>&1 | grep -v "amdgpu.ids" > $O/attn_bench_agpr.txt
cp /tmp/new.so refign_amd/lib/librefign_hip.so
for f in pytest_subset.txt vgpr_form_ab.txt attn_bench_vgpr_form.txt attn_bench_agpr.txt; do echo "== $f"; cat $O/$f; done
