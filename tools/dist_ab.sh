#!/bin/bash
# A/B of bench.py with and without an initialised 1-rank RCCL process group (same box, same process layout)
run() { timeout 500 python bench.py --gpus 1 --steps 12 --warmup 2 --no-cpu --no-roofline 2>&1 | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
echo -n "plain                        : "; run
echo -n "dist                         : "; env $D MASTER_PORT=29561 bash -c "$(declare -f run); run"
echo -n "plain, 4 queues, prio 0      : "; GPU_MAX_HW_QUEUES=4 RFN_SIDE_PRIORITY=0 bash -c "$(declare -f run); run"
echo -n "dist,  4 queues, prio 0      : "; env $D MASTER_PORT=29562 GPU_MAX_HW_QUEUES=4 RFN_SIDE_PRIORITY=0 bash -c "$(declare -f run); run"
echo -n "dist via torchrun            : "; timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29563 bench.py --gpus 1 --steps 12 --warmup 2 --no-cpu --no-roofline 2>&1 | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'
