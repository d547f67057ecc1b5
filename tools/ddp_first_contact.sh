#!/bin/bash
# tools/ddp_first_contact.sh [N=2] [OUT=gpurun_out/first_contact] -- the first minutes on a box with more than one GPU.
# The N > 1 path has only ever run as one rank (RFN_DDP_REHEARSAL=1) and as two gloo ranks on CPU: this runs,
# IN ORDER and each under a wall-clock guard (GUARD seconds, default 180; a step that hangs is killed and the next one still runs),
#   1. the world-N RCCL SyncBatchNorm worker (tools/micro/syncbn_rccl_worker.py): exchanges over torch's group, over a
#      communicator of our own, captured in a hipGraph; the bucketed gradient all-reduce;
#   2. 3 bench steps, RFN_DDP_MODE=torch (the N > 1 default): every exchange through torch.distributed, eager student passes;
#   3. 3 bench steps, RFN_DDP_MODE=direct: RCCL called directly inside the graphed student passes, passes in stream order
#      (two communicators: student, teacher);
#   4. 3 bench steps, RFN_DDP_MODE=direct3: the mixed pass next to the source pass on a third communicator (the one-GPU schedule);
#   5. the one-GPU line for comparison (weak scaling: ms/step should stay put).
# One line per step in OUT/first_contact.txt: PASS / FAIL(rc) / TIMEOUT, ms per step, pairs/s.  Nothing here kills by pattern: every
# launch is a process group of its own under `timeout`.
N=${1:-2}; O=${2:-gpurun_out/first_contact}; GUARD=${GUARD:-180}
export TMPDIR=${TMPDIR:-/tmp} HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p "$O"; R="$O/first_contact.txt"; : > "$R"
python -c 'import torch; print("torch", torch.__version__, "gpus", torch.cuda.device_count())' | tee -a "$R"   # pages the image in, unguarded
have=$(python -c 'import torch; print(torch.cuda.device_count())')
if [ "$have" -lt "$N" ]; then echo "SKIP: $N ranks asked for, $have GPU(s) visible" | tee -a "$R"; [ "$have" -lt 1 ] && exit 0; N=$have; fi
port=29700
launch() {  # launch LABEL LOG ENV... -- CMD...   (torchrun, N ranks)
  local label=$1 log=$2; shift 2
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  port=$((port + 1))
  local t0=$(date +%s.%N)
  env "${envs[@]}" timeout -k 10 "$GUARD" python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 \
      --master-port $port "$@" > "$O/$log" 2>&1
  local rc=$? t1=$(date +%s.%N)
  local wall=$(python -c "print(f'{$t1 - $t0:.0f}')")
  local line=$(grep '"metric"' "$O/$log" | tail -1 | python -c 'import sys, json
try:
    j = json.loads(sys.stdin.read())
    print("%.1f ms/step  %.2f %s  n_gpus %d" % (j["ms_per_step"], j["value"], j["unit"], j["n_gpus"]))
except Exception:
    pass')
  if [ $rc -eq 0 ]; then s=PASS; elif [ $rc -eq 124 ] || [ $rc -eq 137 ]; then s="TIMEOUT(${GUARD}s)"; else s="FAIL(rc=$rc)"; fi
  printf '%-12s %-78s %s  [%ss]\n' "$s" "$label" "$line" "$wall" | tee -a "$R"
  [ $rc -ne 0 ] && tail -5 "$O/$log" | sed 's/^/      | /' >> "$R"
  return $rc
}
launch "1. RCCL SyncBatchNorm worker, $N ranks" worker.log -- tools/micro/syncbn_rccl_worker.py
grep -E '^(PASS|FAIL)  ' "$O/worker.log" | sed 's/^/      /' >> "$R"
B="bench.py --gpus $N --steps 3 --warmup 2 --no-cpu --no-roofline"
launch "2. bench, RFN_DDP_MODE=torch (default: torch.distributed exchanges, eager student)" torch.log RFN_DDP_MODE=torch -- $B
launch "3. bench, RFN_DDP_MODE=direct (direct exchanges in the student graphs, stream order)" direct.log RFN_DDP_MODE=direct -- $B
launch "4. bench, RFN_DDP_MODE=direct3 (mixed pass on its own communicator next to the source pass)" direct3.log RFN_DDP_MODE=direct3 -- $B
t0=$(date +%s); timeout -k 10 "$GUARD" python bench.py --gpus 1 --steps 3 --warmup 2 --no-cpu --no-roofline > "$O/one_gpu.log" 2>&1; rc=$?
printf '%-12s %-78s %s\n' "$([ $rc -eq 0 ] && echo PASS || echo "FAIL(rc=$rc)")" "5. one GPU, no process group" \
  "$(grep '"metric"' "$O/one_gpu.log" | tail -1 | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms\/step/')" | tee -a "$R"
echo "--- $R"; cat "$R"
