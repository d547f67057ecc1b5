mkdir -p gpurun_out/dbg1
timeout 600 python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "diagnostic" 2>&1 | tail -40 > gpurun_out/dbg1/diag.txt
timeout 600 python bench.py --workload uawarpc_align_512x512 --steps 50 --warmup 5 > gpurun_out/dbg1/k2.txt 2>&1
timeout 900 python bench.py --precision fp32 --no-cpu --steps 10 > gpurun_out/dbg1/fp32.txt 2>&1
