export TMPDIR=/tmp
O=gpurun_out/r02k; mkdir -p $O
run() { timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
D="RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
{
echo -n "dist, 4 queues        : "; env $D MASTER_PORT=29561 GPU_MAX_HW_QUEUES=4 bash -c "$(declare -f run); run"
echo -n "dist, 16 queues       : "; env $D MASTER_PORT=29562 GPU_MAX_HW_QUEUES=16 bash -c "$(declare -f run); run"
echo -n "dist, 32 queues       : "; env $D MASTER_PORT=29563 GPU_MAX_HW_QUEUES=32 bash -c "$(declare -f run); run"
echo -n "dist, side prio 0     : "; env $D MASTER_PORT=29564 RFN_SIDE_PRIORITY=0 bash -c "$(declare -f run); run"
echo -n "dist, lazy PG init    : "; env $D MASTER_PORT=29565 RFN_BENCH_LAZY_PG=1 bash -c "$(declare -f run); run"
echo -n "plain, 4 queues       : "; GPU_MAX_HW_QUEUES=4 bash -c "$(declare -f run); run"
echo -n "plain, 16 queues      : "; GPU_MAX_HW_QUEUES=16 bash -c "$(declare -f run); run"
} > $O/dist_ab2.txt 2>&1
echo "== plain" > $O/probe.txt
timeout 300 python tools/micro/stream_probe.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" >> $O/probe.txt
echo "== dist" >> $O/probe.txt
env $D MASTER_PORT=29566 timeout 300 python tools/micro/stream_probe.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" >> $O/probe.txt
