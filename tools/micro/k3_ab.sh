run() { timeout 300 python bench.py --workload refign_daformer_step_1080x1920 --height 512 --width 1024 --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
for cfg in "" "RFN_FUSED_GELU_BWD=0" "RFN_GEMM_NT_MIN_TILES=1000" "RFN_BN_WGS=2048" "RFN_CORR_VARIANT=41" "RFN_FUSED_GELU_BWD=0 RFN_GEMM_NT_MIN_TILES=1000 RFN_BN_WGS=2048 RFN_CORR_VARIANT=41" ""; do
  echo -n "${cfg:-default} : "; env $cfg bash -c "$(declare -f run); run"
done
