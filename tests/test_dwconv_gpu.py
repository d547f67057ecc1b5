"""GPU: hand-written depthwise 3x3 (csrc/dwconv.hip) forward / backward-data / backward-weight against torch's
conv2d (fp32 reference of the same op), fp32 and bf16 activations, dilations 1/6, ragged sizes."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,C,dil", [(2, 7, 9, 32, 1), (1, 17, 30, 256, 1), (2, 33, 41, 64, 6), (1, 5, 3, 8, 1),
                                        (1, 40, 64, 1024, 12)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dwconv_matches_conv2d(dev, B, H, W, C, dil, dtype):
    from refign_amd.dwconv import dwconv3x3_nhwc
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + C + dil)
    x = torch.randn(B, H, W, C, generator=g).to(dev)
    w = (0.3 * torch.randn(C, 1, 3, 3, generator=g)).to(dev).requires_grad_()
    b = (0.1 * torch.randn(C, generator=g)).to(dev).requires_grad_()
    gy = torch.randn(B, H, W, C, generator=g).to(dev)
    xa = x.to(dtype).requires_grad_()
    y = dwconv3x3_nhwc(xa, w, b, dil)
    assert y.dtype == dtype and y.shape == x.shape
    y.backward(gy.to(dtype))
    # fp32 reference on the SAME (possibly bf16-rounded) inputs
    xr = xa.detach().float().permute(0, 3, 1, 2).requires_grad_()
    wr, br = w.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    yr = F.conv2d(xr, wr, br, padding=dil, dilation=dil, groups=C)
    yr.backward(gy.to(dtype).float().permute(0, 3, 1, 2))
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    assert torch.allclose(y.float(), yr.permute(0, 2, 3, 1), **tol)
    assert torch.allclose(xa.grad.float(), xr.grad.permute(0, 2, 3, 1), **tol)
    npix = B * H * W
    wtol = dict(rtol=1e-3, atol=1e-3 * npix ** 0.5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2 * npix ** 0.5)
    assert torch.allclose(w.grad, wr.grad, **wtol)
    assert torch.allclose(b.grad, br.grad, **wtol)


def test_mix_ffn_dwconv_tokens_equals_reference_formulation(dev):
    """DWConv on tokens == transpose -> NCHW depthwise conv -> transpose (mix_transformer.py:563-567)"""
    from refign_amd.seg import DWConv
    m = DWConv(64).to(dev)
    x = torch.randn(2, 12 * 20, 64, device=dev)
    want = m.dwconv(x.transpose(1, 2).reshape(2, 64, 12, 20)).flatten(2).transpose(1, 2)
    assert torch.allclose(m(x, 12, 20), want, rtol=1e-4, atol=1e-5)
