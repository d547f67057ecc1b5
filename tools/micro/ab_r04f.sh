#!/bin/bash
# tools/micro/ab_r04f.sh -- fragment reads ahead of the MFMAs in gemm_nt / gemm_tn3 / attention (refign_amd/lib/ab/librefign_new.so)
# against the library of the commit before (librefign_old.so): parity tests on the new one, then the step alternating, then shapes
cd "$(dirname "$0")/../.."
L=refign_amd/lib/librefign_hip.so
cp refign_amd/lib/ab/librefign_new.so $L
python -m pytest tests/test_mfma_gpu.py tests/test_race_gpu.py tests/test_seg_gpu.py tests/test_step_gpu.py -x -q -m gpu 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
for v in old new old new old new; do cp refign_amd/lib/ab/librefign_$v.so $L; echo -n "$v : "; run; done
for v in old new; do cp refign_amd/lib/ab/librefign_$v.so $L; echo "== $v"; SWEEP_CFGS="x" SWEEP_PERSIST="" python tools/gemm_sweep.py | head -1; SWEEP_CFGS=";" SWEEP_PERSIST=0 python tools/gemm_sweep.py | tail -1; python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | head -30; done
cp refign_amd/lib/ab/librefign_new.so $L
