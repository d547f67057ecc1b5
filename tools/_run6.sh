export TMPDIR=/tmp
O=gpurun_out/r02m; mkdir -p $O
timeout 900 python -m pytest tests/test_syncbn_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_syncbn.txt
run() { timeout 300 python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*"hipgraph_regions": \({[^}]*}\).*/\1 ms  \2/'; }
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
{
echo -n "rehearsal, eager student (N > 1 default), buckets released inside the last backward : "; env $D MASTER_PORT=29562 RFN_DDP_REHEARSAL=1 bash -c "$(declare -f run); run"
echo -n "rehearsal, RFN_GRAPH_DDP=1 (graphed student, mixed pass on its own communicator + stream) : "; env $D MASTER_PORT=29563 RFN_DDP_REHEARSAL=1 RFN_GRAPH_DDP=1 bash -c "$(declare -f run); run"
echo -n "rehearsal, RFN_GRAPH_DDP=1 RFN_MIXED_CONCURRENT=0 : "; env $D MASTER_PORT=29564 RFN_DDP_REHEARSAL=1 RFN_GRAPH_DDP=1 RFN_MIXED_CONCURRENT=0 bash -c "$(declare -f run); run"
} > $O/dist_ab4.txt 2>&1
