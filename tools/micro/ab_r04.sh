run() { timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline 2>/dev/null | grep '"metric"' | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1 ms/'; }
python tools/micro/tn_rect_ab.py
python -m pytest tests/test_mfma_gpu.py -x -q -k "tn or linear or mlp or ffn" 2>&1 | tail -2
echo -n "default                : "; run
echo -n "RFN_GEMM_TN_RECT=0     : "; RFN_GEMM_TN_RECT=0 run
echo -n "RFN_FUSED_GELU_BWD=1   : "; RFN_FUSED_GELU_BWD=1 run
echo -n "default again          : "; run
