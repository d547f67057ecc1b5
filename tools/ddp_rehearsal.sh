#!/bin/bash
# tools/ddp_rehearsal.sh OUT -- what ONE rank of N > 1 does, on the one GPU of a development box: a 1-rank RCCL process
# group with RFN_DDP_REHEARSAL=1 (every SyncBatchNorm exchange issued, teacher communicator, gradient buckets released
# inside the last backward and all-reduced), eager student (the N > 1 default) and graphed student (RFN_GRAPH_DDP=1),
# next to the plain one-GPU step with and without a process group.
export TMPDIR=/tmp
O=gpurun_out/${1:-ddp}; mkdir -p $O
run() { timeout 300 python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
{
echo -n "one GPU, no process group                         : "; run
echo -n "one GPU, 1-rank RCCL group                        : "; env $D MASTER_PORT=29561 bash -c "$(declare -f run); run"
echo -n "rehearsal of one rank of N, eager student (default): "; env $D MASTER_PORT=29562 RFN_DDP_REHEARSAL=1 bash -c "$(declare -f run); run"
echo -n "rehearsal of one rank of N, RFN_GRAPH_DDP=1        : "; env $D MASTER_PORT=29563 RFN_DDP_REHEARSAL=1 RFN_GRAPH_DDP=1 bash -c "$(declare -f run); run"
echo -n "the same under torchrun (1 rank)                  : "; RFN_DDP_REHEARSAL=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29564 bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'
} > $O/ddp_rehearsal.txt 2>&1
