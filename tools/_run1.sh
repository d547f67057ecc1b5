export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
timeout 900 python -m pytest tests/test_syncbn_gpu.py tests/test_step_gpu.py -x -q -m gpu 2>&1 | tail -4 > $O/pytest_subset.txt
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29563 bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline 2>$O/torchrun.err | grep '"metric"' > $O/bench_torchrun_1rank.json
for op in aten::copy_ aten::cat aten::add aten::mul aten::add_; do
  echo "== $op" >> $O/stacks.txt
  RFN_GRAPH_STUDENT=0 timeout 300 python tools/step_profile.py --stacks $op --rows 30 2>&1 | grep "^n=" >> $O/stacks.txt
done
