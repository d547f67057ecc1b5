mkdir -p gpurun_out/corrbwd
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "corr" 2>&1 | tail -8 > gpurun_out/corrbwd/tests.txt
timeout 300 python tools/kbench.py --only L1,L2,K2-L1 2>&1 | grep "backward" > gpurun_out/corrbwd/kbench.txt
timeout 300 python tools/matcher_bench.py --precision fp16 2>&1 | tail -2 >> gpurun_out/corrbwd/kbench.txt
