# rehearsal of one rank of N (torch mode) and the plain run with 4 (default) and 8 hardware queues per process
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config'].get('mixed_pass_stream_probe'))"; }
p=29800
for q in 4 8 4 8; do
 echo "== plain, GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 300 python $R/bench.py --no-cpu --no-roofline --steps 10 --warmup 5 2>/dev/null | tail -1 | one
 p=$((p+1))
 echo "== rehearsal torch, GPU_MAX_HW_QUEUES=$q"
 GPU_MAX_HW_QUEUES=$q RFN_DDP_REHEARSAL=1 RFN_DDP_MODE=torch timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $p $R/bench.py --gpus 1 --no-cpu --no-roofline --steps 10 --warmup 5 2>/dev/null | grep '^{"metric"' | one
done
