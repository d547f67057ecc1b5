export TMPDIR=/tmp
O=gpurun_out/r02l; mkdir -p $O
run() { timeout 300 python bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*"hipgraph_regions": \({[^}]*}\).*/\1 ms  \2/'; }
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
{
echo -n "plain                                : "; run
echo -n "1-rank RCCL group                    : "; env $D MASTER_PORT=29561 bash -c "$(declare -f run); run"
echo -n "1-rank RCCL group, RFN_DDP_REHEARSAL : "; env $D MASTER_PORT=29562 RFN_DDP_REHEARSAL=1 bash -c "$(declare -f run); run"
echo -n "  + RFN_GRAPH_DDP=0 (eager student)  : "; env $D MASTER_PORT=29563 RFN_DDP_REHEARSAL=1 RFN_GRAPH_DDP=0 bash -c "$(declare -f run); run"
echo -n "torchrun, 1 rank, rehearsal          : "; RFN_DDP_REHEARSAL=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29564 bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'
} > $O/dist_ab3.txt 2>&1
env $D MASTER_PORT=29566 RFN_DDP_REHEARSAL=1 timeout 300 python tools/opt_phase_debug.py 2>&1 | grep "^step" > $O/phases_rehearsal.txt
env $D MASTER_PORT=29567 RFN_DDP_REHEARSAL=1 timeout 300 python tools/overlap_debug.py 2>&1 | grep "step\|host" >> $O/phases_rehearsal.txt
timeout 900 python -m pytest tests/test_syncbn_gpu.py tests/test_step_gpu.py -x -q -m gpu 2>&1 | tail -3 > $O/pytest_subset.txt
