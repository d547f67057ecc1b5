"""tools/micro/gemm2_256_ab.py -- isolated timing of the N % 256 == 0 shapes of the teacher on the shipped gemm_nt dispatch
(run twice: RFN_GEMM2=1 -> first-generation 256 x 256 8-wave tile, default -> second-generation 192 x 256 tiles).
Operands rotate over 6 sets so that they come from HBM, 20 launches per replayed hipGraph."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from refign_amd.mfma import gemm_nt  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [  # M, N, K, bias, res      (teacher, 40 views)
    (1296000, 256, 1024, False, False),   # decode head: ASPP 1x1 branches
    (326400, 512, 128, True, False),      # stage 2 fc1
    (326400, 128, 512, True, True),       # stage 2 fc2 (N = 128: not a 256 shape, control)
    (326400, 256, 128, False, False),     # stage 2 kv (reduced tokens would be smaller; upper bound)
    (20400, 2048, 512, True, False),      # stage 4 fc1
    (20400, 512, 2048, True, True),       # stage 4 fc2
    (20400, 512, 512, True, True),        # stage 4 proj
    (20400, 1024, 512, False, False),     # stage 4 kv
    (1305600, 256, 64, True, False),      # stage 1 fc1 (K = 64: stays on the first generation)
]
print("RFN_GEMM2 =", os.environ.get("RFN_GEMM2", "(default 2)"))
for M, N, K, bias, res in SHAPES:
    R = 6 if M * (N + K) * 2 < 1.5e9 else 3
    xs = [torch.randn(M, K, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5 for _ in range(R)]
    b = torch.randn(N, device=dev, dtype=torch.bfloat16) if bias else None
    rs = [torch.randn(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)] if res else None
    for i in range(R):
        gemm_nt(xs[i], ws[i], b, res=None if rs is None else rs[i])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for i in range(18):
                gemm_nt(xs[i % R], ws[i % R], b, res=None if rs is None else rs[i % R])
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 54
    print(f"{M:8d} x {K:4d} -> {N:4d} {'+b' if bias else '  '}{'+res' if res else '    '}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s")
    del xs, ws, rs
