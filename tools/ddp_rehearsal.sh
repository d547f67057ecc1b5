#!/bin/bash
# tools/ddp_rehearsal.sh -- what ONE rank of N costs, on one GPU: torchrun with one rank and RFN_DDP_REHEARSAL=1 (a one-rank group
# issues every collective a rank of N issues), for each exchange mode; bench.py --steps 10 --warmup 5, same box.
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config'].get('hipgraph_regions'))"; }
echo "== plain N = 1"; timeout 300 python $R/bench.py --no-cpu --no-roofline --steps 10 --warmup 5 2>/dev/null | tail -1 | one
p=29650
for m in "torch 0" "torch 1" direct direct3; do
  set -- $m; p=$((p+1))
  echo "== torchrun 1 rank, RFN_DDP_REHEARSAL=1 RFN_DDP_MODE=$1 ${2:+RFN_GRAPH_SEGMENTS=$2}"
  RFN_DDP_REHEARSAL=1 RFN_DDP_MODE=$1 RFN_GRAPH_SEGMENTS=${2:-1} timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port $p $R/bench.py --gpus 1 --no-cpu --no-roofline --steps 10 --warmup 5 2>/dev/null | grep '^{"metric"' | one
done
