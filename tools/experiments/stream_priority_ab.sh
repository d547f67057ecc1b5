# tools/experiments/stream_priority_ab.sh -- priority of the teacher branch's stream (uda._SIDE_PRIORITY: -1 = rounds 2-5's high-priority
# stream, 0 = default priority, probed for concurrency) in the headline configuration, with adapt_to_ref, and in the one-rank
# rehearsals of the data-parallel modes (profiles/r06_stream_priority_ab.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config'].get('mixed_pass_stream_probe'))"; }
p=29700
for pr in -1 0; do
 echo "== plain, side priority $pr"; timeout 300 python $R/tools/ab_const.py uda._SIDE_PRIORITY=$pr -- --no-cpu --no-roofline --steps 10 --warmup 5 2>/dev/null | tail -1 | one
 echo "== adapt, side priority $pr"; timeout 300 python $R/tools/ab_const.py uda._SIDE_PRIORITY=$pr -- --adapt-to-ref --no-cpu --no-roofline --steps 20 --warmup 5 2>/dev/null | tail -1 | one
 for m in torch direct3; do p=$((p+1))
  echo "== rehearsal $m, side priority $pr"
  RFN_DDP_REHEARSAL=1 RFN_DDP_MODE=$m timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $p $R/tools/ab_const.py uda._SIDE_PRIORITY=$pr -- --gpus 1 --no-cpu --no-roofline --steps 10 --warmup 5 2>/dev/null | grep '^{"metric"' | one
 done
done
