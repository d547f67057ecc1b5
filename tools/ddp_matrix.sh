#!/bin/bash
# tools/ddp_matrix.sh OUT -- the rehearsed rank's step under the gradient-reduce / mixed-pass choices, each twice (run-to-run spread)
export TMPDIR=/tmp
O=gpurun_out/${1:-ddpm}; mkdir -p $O
run() { timeout 300 python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 RFN_DDP_REHEARSAL=1"
port=29580
row() { local label=$1; shift; for k in 1 2; do port=$((port+1)); echo -n "$label : "; env $D MASTER_PORT=$port "$@" bash -c "$(declare -f run); run"; done; }
{
row "mixed pass concurrent, tail reduce over torch.distributed (default)        "
row "mixed pass concurrent, tail reduce over our own communicator               " RFN_DDP_DIRECT_REDUCE=1
row "mixed pass concurrent, two buffers reduced separately (in-graph releases)  " RFN_DDP_DIRECT_REDUCE=1 RFN_DDP_TWO_BUFFER=1
row "stream order, in-graph releases over our own communicator                  " RFN_DDP_MIXED_COMM=0 RFN_DDP_DIRECT_REDUCE=1
row "stream order, tail reduce over torch.distributed                           " RFN_DDP_MIXED_COMM=0
} > $O/ddp_matrix.txt 2>&1
cat $O/ddp_matrix.txt
