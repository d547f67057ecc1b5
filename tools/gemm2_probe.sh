#!/bin/bash
# tools/gemm2_probe.sh -- build and run tools/micro/gemm2_probe.hip on the GPU box (gpurun -- bash tools/gemm2_probe.sh)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/gemm2_probe.hip -o /tmp/gemm2_probe -ldl
timeout 900 /tmp/gemm2_probe refign_amd/lib/librefign_hip.so 2>&1 | tee gpurun_out/gemm2_probe.txt
