#!/bin/bash
# tools/ddp_pg_cost.sh OUT -- what the mere presence of a 1-rank RCCL process group costs the one-GPU step, piece by piece
export TMPDIR=/tmp
O=gpurun_out/${1:-pgcost}; mkdir -p $O
run() { timeout 300 python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
{
echo -n "no process group                                     : "; run
echo -n "1-rank group                                         : "; env $D MASTER_PORT=29571 bash -c "$(declare -f run); run"
echo -n "1-rank group, no SyncBatchNorm conversion            : "; env $D MASTER_PORT=29572 RFN_BENCH_SYNC_BN=0 bash -c "$(declare -f run); run"
echo -n "1-rank group, communicator created lazily            : "; env $D MASTER_PORT=29573 RFN_BENCH_LAZY_PG=1 bash -c "$(declare -f run); run"
echo -n "1-rank group, no gradient all-reduce (RFN_DDP_SKIP_REDUCE=1) : "; env $D MASTER_PORT=29574 RFN_DDP_SKIP_REDUCE=1 bash -c "$(declare -f run); run"
echo -n "1-rank group, lazy + no conversion + no reduce       : "; env $D MASTER_PORT=29575 RFN_BENCH_LAZY_PG=1 RFN_BENCH_SYNC_BN=0 RFN_DDP_SKIP_REDUCE=1 bash -c "$(declare -f run); run"
} > $O/pg_cost.txt 2>&1
cat $O/pg_cost.txt
