R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config'].get('mixed_pass_stream_probe'))"; }
p=29700
for pr in -1 0; do
 echo "== plain, side priority $pr"; RFN_SIDE_PRIORITY=$pr timeout 300 python $R/bench.py --no-cpu --no-roofline --steps 10 --warmup 5 2>/dev/null | tail -1 | one
 echo "== adapt, side priority $pr"; RFN_SIDE_PRIORITY=$pr timeout 300 python $R/bench.py --adapt-to-ref --no-cpu --no-roofline --steps 20 --warmup 5 2>/dev/null | tail -1 | one
 for m in torch direct3; do p=$((p+1))
  echo "== rehearsal $m, side priority $pr"
  RFN_SIDE_PRIORITY=$pr RFN_DDP_REHEARSAL=1 RFN_DDP_MODE=$m timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $p $R/bench.py --gpus 1 --no-cpu --no-roofline --steps 10 --warmup 5 2>/dev/null | grep '^{"metric"' | one
 done
done
