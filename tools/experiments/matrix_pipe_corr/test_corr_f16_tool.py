"""The matrix-pipe local correlation layer (csrc/corr_f16.hip) and its operand producer (warp.hip split_f16_kernel) against
the numpy oracle of LocalFeatureCorrelationLayer (oracle/cpu_oracle.py, restating models/modules.py:266-274 over
correlation.cpp:80-129) and against the fp32 VALU kernel.  Tolerance: the split operands carry 22 significand bits, the
products are accumulated in fp32 in a different order -- 2e-6 absolute on unit-norm features (the fp32 kernel vs the fp64
oracle: 1e-6)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests"))

pytestmark = pytest.mark.gpu


def _feat(B, C, H, W, seed, dev):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(B, C, H, W, generator=g), dim=1).to(dev)


def test_split_f16_layout_and_precision():
    from corr_experiments import split_f16
    dev = torch.device("cuda:0")
    x = _feat(2, 64, 19, 37, 1, dev) * 3.0
    s = split_f16(x)
    assert s.shape == (2, 2, 2, 19, 37, 32)
    hi = x.half()
    lo = (x - hi.float()).half()
    want = torch.stack([hi, lo], 1).view(2, 2, 2, 32, 19, 37).permute(0, 2, 1, 4, 5, 3)      # (b, part, chunk, c, y, x) -> ...
    assert torch.equal(s, want.contiguous())
    rec = (s[:, :, 0].float() + s[:, :, 1].float()).permute(0, 1, 4, 2, 3).reshape(2, 64, 19, 37)
    assert float((rec - x).abs().max()) <= 2.0 ** -21 * float(x.abs().max())


@pytest.mark.parametrize("B,C,H,W", [(1, 32, 8, 32), (2, 64, 19, 37), (1, 128, 33, 70), (2, 256, 16, 24), (1, 128, 9, 100)])
@pytest.mark.parametrize("fuse", [True, False])
def test_corr_f16_matches_oracle_and_fp32_kernel(B, C, H, W, fuse, oracle):
    import corr_experiments as corr
    dev = torch.device("cuda:0")
    src, trg = _feat(B, C, H, W, 2, dev), _feat(B, C, H, W, 3, dev)
    got = corr.local_correlation_layer_split(corr.split_f16(src), corr.split_f16(trg), fuse=fuse).cpu().numpy()
    if fuse:
        want = oracle.local_correlation_layer(src.cpu().numpy(), trg.cpu().numpy())
        ref = corr.local_correlation_layer(src, trg).cpu().numpy()
    else:
        want = oracle.corr_forward(trg.cpu().numpy(), src.cpu().numpy(), patch_size=9).reshape(B, 81, H, W)
        ref = corr.spatial_correlation_sample(trg, src, patch_size=9).reshape(B, 81, H, W).cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-6, np.abs(got - want).max()
    assert np.abs(got - ref).max() <= 2e-6


def test_corr_f16_with_fused_warp_matches_warp_then_correlate(oracle):
    import corr_experiments as corr
    dev = torch.device("cuda:0")
    B, C, H, W = 2, 64, 24, 40
    src, trg = _feat(B, C, H, W, 4, dev), _feat(B, C, H, W, 5, dev)
    g = torch.Generator(device="cpu").manual_seed(6)
    flow = (3.0 * torch.randn(B, 2, H, W, generator=g)).to(dev)
    got = corr.local_correlation_layer_split(corr.split_f16(src, flow), corr.split_f16(trg)).cpu().numpy()
    want = oracle.local_correlation_layer(oracle.warp(src.cpu().numpy(), flow.cpu().numpy()), trg.cpu().numpy())
    assert np.abs(got - want).max() <= 2e-5          # (the warp's own fp32 interpolation: same bound as the fp32 path)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["0", "2"])
def test_corr9_fp32_matrix_pipe_matches_fp64_formulation(cfg, monkeypatch):
    """corr_mfma.hip (wave-private rings = cfg 0, workgroup-shared tiles = cfg 2) against a plain fp64 torch formulation on
    ragged / tiny / multi-tile shapes, raw and with the fused ReLU + L2 norm (1e-5 relative).  The configuration is read once
    per process: a subprocess per cfg."""
    import subprocess
    code = (
        "import os, sys, torch; sys.path.insert(0, %r); import corr_experiments as ce\n"
        "import torch.nn.functional as F\n"
        "dev = torch.device('cuda:0'); g = torch.Generator().manual_seed(1); worst = 0.0\n"
        "for (B, C, H, W) in [(1, 8, 16, 32), (2, 32, 37, 52), (1, 128, 48, 96)]:\n"
        "    a = F.normalize(torch.randn(B, C, H, W, generator=g), dim=1).to(dev); b = F.normalize(torch.randn(B, C, H, W, generator=g), dim=1).to(dev)\n"
        "    for fuse in (False, True):\n"
        "        got = ce.corr9_mfma(a, b, fuse)\n"
        "        if got is None: continue\n"
        "        bp = F.pad(b.double(), (4, 4, 4, 4)); ref = torch.stack([(a.double() * bp[:, :, dy:dy + H, dx:dx + W]).sum(1) for dy in range(9) for dx in range(9)], 1)\n"
        "        if fuse: ref = F.normalize(F.relu(ref), dim=1, eps=1e-12)\n"
        "        worst = max(worst, float((got.double() - ref).abs().max() / ref.abs().max()))\n"
        "print('worst', worst); assert worst < 1e-5\n") % os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RFN_CORR_MFMA_CFG=cfg), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
