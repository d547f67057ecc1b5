export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
run() { timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
{
echo -n "plain                      : "; run
echo -n "dist                       : "; env $D MASTER_PORT=29561 bash -c "$(declare -f run); run"
echo -n "dist, mixed serial         : "; env $D MASTER_PORT=29562 RFN_MIXED_CONCURRENT=0 bash -c "$(declare -f run); run"
echo -n "dist, no prefetch          : "; env $D MASTER_PORT=29563 RFN_PREFETCH_NEXT=0 bash -c "$(declare -f run); run"
echo -n "dist, no SyncBN conversion : "; env $D MASTER_PORT=29564 RFN_BENCH_SYNC_BN=0 bash -c "$(declare -f run); run"
echo -n "dist, torch AdamW          : "; env $D MASTER_PORT=29565 RFN_ADAMW_KERNEL=0 bash -c "$(declare -f run); run"
} > $O/dist_ab.txt 2>&1
env $D MASTER_PORT=29566 timeout 300 python tools/opt_phase_debug.py 2>&1 | grep "^step" > $O/phases_dist.txt
env $D MASTER_PORT=29567 timeout 300 python tools/overlap_debug.py 2>&1 | grep "step\|host" >> $O/phases_dist.txt
RFN_HIP_GRAPH=0 timeout 300 python tools/step_profile.py --stacks aten::copy_ --rows 40 2>&1 | grep "^n=" > $O/stacks_copy_eager_all.txt
