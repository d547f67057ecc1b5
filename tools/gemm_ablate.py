"""tools/gemm_ablate.py -- where a launch of the NT GEMM kernel spends its time: the same shapes with the DMA, the MFMAs or the
stores switched off (RFN_GEMM_ABLATE bits 1 / 2 / 4; results are then meaningless).  Needs a PROFILING build of the library:
`make -C refign_amd/csrc clean all FLAGS+=" -DRFN_GEMM_PROFILE"` (the product build has no such hooks).  One process per setting.
Shapes below 20 us are bounded by this harness's host launch rate (~10 us per call), not by the kernel."""
import os
import subprocess
import sys

SHAPES = [(8160, 320, 320), (8160, 1280, 320), (8160, 320, 1280), (81600, 320, 320), (81600, 1280, 320), (81600, 320, 1280),
          (129600, 256, 64), (1296000, 64, 64), (326400, 512, 128)]
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from refign_amd import mfma
dev = torch.device("cuda:0")
def timeit(fn, reps=30):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
out = []
for M, N, K in %r:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    out.append(timeit(lambda: mfma.gemm_nt(x, w, b)))
print(" ".join("%%.1f" %% v for v in out))
'''

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    print("ablate  " + "  ".join(f"{m}x{k}>{n}" for m, n, k in SHAPES))
    for a in (0, 1, 2, 4, 3, 5, 6, 7):
        env = dict(os.environ, RFN_GEMM_ABLATE=str(a))
        r = subprocess.run([sys.executable, "-c", CHILD % (root, SHAPES)], env=env, capture_output=True, text=True)
        print(f"{a:6d}  " + r.stdout.strip() + (r.stderr[-300:] if r.returncode else ""), flush=True)
