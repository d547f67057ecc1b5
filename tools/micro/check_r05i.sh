#!/bin/bash
# round 5, ninth call: mfma_gemm.hip with MFMA results in VGPRs (gemm2 in its own translation unit) -- whole GPU suite, step A/B
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05i; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu --tb=short 2>&1 | tail -6 > $O/pytest_gpu.txt
run() { timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>&1 | grep '^{"metric"' | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['ms_per_step'])"; }
cp refign_amd/lib/librefign_hip.so /tmp/new.so
{
for i in 1 2 3; do
cp /tmp/new.so refign_amd/lib/librefign_hip.so
echo -n "mfma_gemm.hip: MFMA results in VGPRs       : "; run
cp refign_amd/lib/ab/librefign_hip_noflag.so refign_amd/lib/librefign_hip.so
echo -n "mfma_gemm.hip: compiler's choice (AGPRs)   : "; run
done
cp /tmp/new.so refign_amd/lib/librefign_hip.so
} > $O/gemm_vgpr_form_ab.txt 2>&1
cp /tmp/new.so refign_amd/lib/librefign_hip.so
SWEEP_CFGS=";64,64,2" SWEEP_PERSIST=0 timeout 300 python tools/gemm_sweep.py 2>&1 | grep -v "amdgpu.ids" | head -40 > $O/gemm_sweep_vgpr.txt
for f in pytest_gpu.txt gemm_vgpr_form_ab.txt gemm_sweep_vgpr.txt; do echo "== $f"; cat $O/$f; done
