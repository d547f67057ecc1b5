"""halo-tiled 3x3 convolution (csrc/conv3x3.hip) against torch fp32 on the operands' 16-bit values, and its time against the
implicit-GEMM kernel (RFN_CONV_HALO=0 in another process).  python tools/experiments/conv3x3_check.py [time]"""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from refign_amd import conv
dev = torch.device("cuda:0")
torch.manual_seed(0)
cases = [  # B, H, W, C, N, dtype, act
    (2, 37, 61, 64, 64, torch.float16, 'relu'),
    (1, 16, 32, 128, 128, torch.float16, 'relu'),
    (2, 135, 240, 256, 256, torch.bfloat16, None),
    (3, 40, 70, 128, 64, torch.bfloat16, 'leaky'),
    (1, 9, 200, 64, 192, torch.float16, None),
]
if "time" in sys.argv:
    cases = [(4, 1080, 1920, 64, 64, torch.float16, 'relu'), (4, 540, 960, 64, 128, torch.float16, 'relu'), (4, 540, 960, 128, 128, torch.float16, 'relu'),
             (4, 270, 480, 128, 256, torch.float16, 'relu'), (4, 270, 480, 256, 256, torch.float16, 'relu'), (4, 135, 240, 256, 512, torch.float16, 'relu'),
             (4, 135, 240, 512, 512, torch.float16, 'relu'), (44, 135, 240, 1024, 256, torch.bfloat16, None), (4, 135, 240, 1024, 256, torch.bfloat16, None)]
for B, H, W, C, N, dt, act in cases:
    x = torch.randn(B, C, H, W, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(N, C, 3, 3, device=dev) * (9 * C) ** -0.5)
    bias = torch.randn(N, device=dev) * 0.1
    with torch.no_grad():
        y = conv.conv2d_mfma(x, w, bias, 1, 1, 1, act=act, dtype=dt)
        assert y is not None
        if "time" in sys.argv:
            for _ in range(3):
                conv.conv2d_mfma(x, w, bias, 1, 1, 1, act=act, dtype=dt)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                conv.conv2d_mfma(x, w, bias, 1, 1, 1, act=act, dtype=dt)
            torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 10 * 1e6
            fl = 2.0 * B * H * W * N * 9 * C
            print(f"{B}x{H}x{W} C={C} N={N} {str(dt)[6:]}: {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  ({fl / us / 1e6 / 2500:.3f} of the MFMA peak)", flush=True)
            continue
        ref = F.conv2d(x.float(), w.to(dt).float(), bias.to(dt).float(), padding=1)
        if act == 'relu':
            ref = F.relu(ref)
        elif act == 'leaky':
            ref = F.leaky_relu(ref, 0.1)
    err = float((y.float() - ref).abs().max()) / float(ref.abs().max())
    print(f"{B}x{H}x{W} C={C} N={N} {str(dt)[6:]} {act}: max err / max |ref| = {err:.2e}", flush=True)
    assert err < (1.2e-2 if dt == torch.bfloat16 else 2e-3), err
print("ok")
