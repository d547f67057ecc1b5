#!/bin/bash
# tools/ddp_rehearsal.sh OUT -- what ONE rank of N > 1 does, on the one GPU of a development box: a 1-rank RCCL process
# group with RFN_DDP_REHEARSAL=1 (every SyncBatchNorm exchange issued, teacher communicator, gradient buckets released
# inside the last backward and all-reduced): the N > 1 defaults (exchanges as RCCL calls of our own, graphed student passes,
# mixed pass next to the source pass) and the alternatives, next to the plain one-GPU step with and without a process group.
export TMPDIR=/tmp
O=gpurun_out/${1:-ddp}; mkdir -p $O
run() { timeout 300 python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
{
echo -n "one GPU, no process group                                        : "; run
echo -n "one GPU, 1-rank RCCL group                                       : "; env $D MASTER_PORT=29561 bash -c "$(declare -f run); run"
echo -n "one rank of N, defaults (direct RCCL exchanges, graphed student) : "; env $D MASTER_PORT=29562 RFN_DDP_REHEARSAL=1 bash -c "$(declare -f run); run"
echo -n "  the guard's second attempt (fake stall after the first step, RFN_BENCH_STALL_S=25)  : "; RFN_DDP_REHEARSAL=1 RFN_BENCH_FAKE_STALL=1 RFN_BENCH_STALL_S=25 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29569 bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline 2>$O/second_attempt.err | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*"attempt": \([0-9]\).*/\1 ms (attempt \2)/'
echo -n "  the guard's second attempt (an exception in the first set-up step)               : "; RFN_DDP_REHEARSAL=1 RFN_BENCH_FAKE_RAISE=1 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29570 bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline 2>$O/second_attempt_raise.err | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*"attempt": \([0-9]\).*/\1 ms (attempt \2)/'
echo -n "  gradient reduce over a communicator of our own at the tail (RFN_DDP_DIRECT_REDUCE=1) : "; env $D MASTER_PORT=29567 RFN_DDP_REHEARSAL=1 RFN_DDP_DIRECT_REDUCE=1 bash -c "$(declare -f run); run"
echo -n "  stream order + in-graph gradient releases (RFN_DDP_MIXED_COMM=0 RFN_DDP_DIRECT_REDUCE=1) : "; env $D MASTER_PORT=29568 RFN_DDP_REHEARSAL=1 RFN_DDP_MIXED_COMM=0 RFN_DDP_DIRECT_REDUCE=1 bash -c "$(declare -f run); run"
echo -n "  direct RCCL exchanges, eager student (RFN_GRAPH_DDP=0)         : "; env $D MASTER_PORT=29563 RFN_DDP_REHEARSAL=1 RFN_GRAPH_DDP=0 bash -c "$(declare -f run); run"
echo -n "  torch.distributed exchanges, graphed (RFN_RCCL_DIRECT=0 RFN_GRAPH_DDP=1) : "; env $D MASTER_PORT=29564 RFN_DDP_REHEARSAL=1 RFN_RCCL_DIRECT=0 RFN_GRAPH_DDP=1 bash -c "$(declare -f run); run"
echo -n "  torch.distributed exchanges, eager (RFN_RCCL_DIRECT=0)         : "; env $D MASTER_PORT=29565 RFN_DDP_REHEARSAL=1 RFN_RCCL_DIRECT=0 bash -c "$(declare -f run); run"
echo -n "one rank of N, defaults, under torchrun (1 rank)                 : "; RFN_DDP_REHEARSAL=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29566 bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'
} > $O/ddp_rehearsal.txt 2>&1
