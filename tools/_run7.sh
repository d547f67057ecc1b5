export TMPDIR=/tmp
O=gpurun_out/r02m; mkdir -p $O
timeout 900 python -m pytest tests/test_syncbn_gpu.py -x -q -m gpu -k rccl 2>&1 | grep -v "^$" | tail -60 > $O/pytest_syncbn2.txt
