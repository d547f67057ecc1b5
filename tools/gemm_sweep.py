#!/usr/bin/env python3
"""tools/gemm_sweep.py -- rfn_gemm_nt tile / ring-depth sweep (RFN_GEMM_CFG is read once per process: one subprocess per
configuration) on the Linear shapes of the Refign step."""
import os
import subprocess
import sys

SHAPES = [(8160, 320, 320), (8160, 1280, 320), (8160, 320, 1280), (8160, 640, 320), (2040, 320, 1280), (2040, 1280, 320),
          (32640, 128, 128), (32640, 512, 128), (32640, 128, 512), (129600, 64, 64), (129600, 256, 64), (129600, 64, 256),
          (2040, 2048, 512), (81600, 320, 320), (81600, 1280, 320), (81600, 320, 1280),
          (20400, 512, 2048), (20400, 2048, 512), (326400, 512, 128), (1296000, 256, 64), (1296000, 256, 1024)]
CFGS = ["", "128,128,2", "128,64,2", "64,128,2", "64,64,2", "256,256,2", "128,128,3", "128,64,3", "64,64,4", "64,64,8"]
if os.environ.get("SWEEP_CFGS"):
    CFGS = os.environ["SWEEP_CFGS"].split(";")

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from refign_amd import mfma
    dev = torch.device("cuda:0")
    out = []
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = torch.randn(N, K, device=dev).bfloat16()
        b = torch.randn(N, device=dev).bfloat16()
        try:
            # a replayed hipGraph of 20 launches: no host launch gaps (the small shapes are otherwise host-bound)
            y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            for _ in range(3):
                mfma.gemm_nt(x, w, b, out=y)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _ in range(20):
                    mfma.gemm_nt(x, w, b, out=y)
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 10)
        except RuntimeError:
            out.append(float("nan"))
    print(" ".join(f"{t:8.1f}" for t in out))
else:
    print(f"{'cfg (bm,bn,ring) / us':24s}" + " ".join(f"{M}x{K}>{N}".rjust(16) for M, N, K in SHAPES))
    for persist in os.environ.get("SWEEP_PERSIST", "0,1").split(","):
        for cfg in CFGS:
            env = dict(os.environ, RFN_GEMM_PERSIST=persist)
            if cfg:
                env["RFN_GEMM_CFG"] = cfg
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "failed: " + r.stderr[-200:]
            vals = line.split()
            print(f"p={persist} {cfg or 'heuristic':18s} " + " ".join(v.rjust(16) for v in vals), flush=True)
