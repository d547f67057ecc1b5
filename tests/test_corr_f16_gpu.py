"""The matrix-pipe local correlation layer (csrc/corr_f16.hip) and its operand producer (warp.hip split_f16_kernel) against
the numpy oracle of LocalFeatureCorrelationLayer (oracle/cpu_oracle.py, restating models/modules.py:266-274 over
correlation.cpp:80-129) and against the fp32 VALU kernel.  Tolerance: the split operands carry 22 significand bits, the
products are accumulated in fp32 in a different order -- 2e-6 absolute on unit-norm features (the fp32 kernel vs the fp64
oracle: 1e-6)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _feat(B, C, H, W, seed, dev):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(B, C, H, W, generator=g), dim=1).to(dev)


def test_split_f16_layout_and_precision():
    from refign_amd.correlation import split_f16
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
    from refign_amd import correlation as corr
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
    from refign_amd import correlation as corr
    dev = torch.device("cuda:0")
    B, C, H, W = 2, 64, 24, 40
    src, trg = _feat(B, C, H, W, 4, dev), _feat(B, C, H, W, 5, dev)
    g = torch.Generator(device="cpu").manual_seed(6)
    flow = (3.0 * torch.randn(B, 2, H, W, generator=g)).to(dev)
    got = corr.local_correlation_layer_split(corr.split_f16(src, flow), corr.split_f16(trg)).cpu().numpy()
    want = oracle.local_correlation_layer(oracle.warp(src.cpu().numpy(), flow.cpu().numpy()), trg.cpu().numpy())
    assert np.abs(got - want).max() <= 2e-5          # (the warp's own fp32 interpolation: same bound as the fp32 path)
