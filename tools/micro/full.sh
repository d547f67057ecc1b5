mkdir -p gpurun_out/full
RFN_TEST_REPORT_DIR=$PWD/gpurun_out/full timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/full/pytest_gpu.txt
python tools/micro/l3.py 2>&1 | grep -v amdgpu.ids > gpurun_out/full/l3.txt
timeout 600 python bench.py --no-cpu 2>/dev/null | tail -1 > gpurun_out/full/bench.json
timeout 600 python bench.py --workload uawarpc_align_512x512 --steps 50 --warmup 5 --no-cpu 2>/dev/null | tail -1 > gpurun_out/full/bench_k2.json
