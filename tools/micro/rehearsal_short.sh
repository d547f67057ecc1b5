#!/bin/bash
# tools/micro/rehearsal_short.sh -- the first rows of tools/ddp_rehearsal.sh (what one rank of N does, on one GPU) + the K3 workload as one rank of N
run() { timeout 300 python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline $EXTRA 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
echo -n "one GPU, no process group                                        : "; run
echo -n "one GPU, 1-rank RCCL group                                       : "; env $D MASTER_PORT=29561 bash -c "$(declare -f run); run"
echo -n "one rank of N, defaults (direct RCCL exchanges, graphed student) : "; env $D MASTER_PORT=29562 RFN_DDP_REHEARSAL=1 bash -c "$(declare -f run); run"
echo -n "one rank of N, defaults, under torchrun (1 rank)                 : "; RFN_DDP_REHEARSAL=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29566 bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'
export EXTRA="--workload refign_daformer_step_1080x1920 --height 512 --width 1024"
echo -n "K3 (DAFormer 512 x 1024), no process group                       : "; run
echo -n "K3, one rank of N, defaults                                      : "; env $D MASTER_PORT=29571 RFN_DDP_REHEARSAL=1 EXTRA="$EXTRA" bash -c "$(declare -f run); run"
