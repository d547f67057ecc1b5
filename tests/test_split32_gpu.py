"""fp32 parity mode on the hand-written kernels (refign_amd/split32.py): split-bf16 products against fp64 formulations.
Bound: 2^-15 relative to the result's scale (two bf16 terms per operand keep 16 significand bits; fp32 itself: 2^-24)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2.0 ** -15


def _r(shape, seed, scale=1.0, dev="cuda:0"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


@pytest.mark.parametrize("M,N,K", [(300, 64, 64), (1000, 19, 256), (513, 320, 320), (77, 8, 100), (2040, 512, 2048)])
def test_linear_fp32_split_forward_backward(M, N, K):
    from refign_amd import split32
    x, w, b = _r((M, K), 1).requires_grad_(True), _r((N, K), 2, K ** -0.5).requires_grad_(True), _r((N,), 3).requires_grad_(True)
    y = split32.linear(x, w, b)
    gy = _r((M, N), 4)
    y.backward(gy)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yd = F.linear(xd, wd, bd)
    yd.backward(gy.double())
    for got, want in ((y, yd), (x.grad, xd.grad), (w.grad, wd.grad), (b.grad, bd.grad)):
        assert float((got.double() - want).abs().max()) <= TOL * float(want.abs().max()) + 1e-6


CONVS = [  # B, H, W, C, N, k, stride, pad, dil
    (2, 17, 23, 64, 64, 3, 1, 1, 1), (1, 33, 40, 3, 64, 7, 4, 3, 1), (2, 20, 28, 64, 128, 3, 2, 1, 1),
    (1, 24, 24, 32, 128, 3, 1, 4, 4), (1, 19, 27, 84, 128, 3, 1, 1, 1), (2, 16, 30, 320, 320, 2, 2, 0, 1),
    (1, 30, 30, 256, 19, 1, 1, 0, 1), (1, 21, 19, 16, 6, 3, 1, 0, 1), (1, 18, 18, 32, 2, 1, 1, 0, 1),
]


@pytest.mark.parametrize("B,H,W,C,N,k,stride,pad,dil", CONVS)
def test_conv2d_fp32_split_forward_backward(B, H, W, C, N, k, stride, pad, dil):
    from refign_amd import split32
    x = _r((B, C, H, W), 5).requires_grad_(True)
    w = _r((N, C, k, k), 6, (C * k * k) ** -0.5).requires_grad_(True)
    b = _r((N,), 7).requires_grad_(True)
    y = split32.conv2d(x, w, b, stride, pad, dil)
    assert y is not None
    gy = _r(tuple(y.shape), 8)
    y.backward(gy)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yd = F.conv2d(xd, wd, bd, stride, pad, dil)
    yd.backward(gy.double())
    for got, want in ((y, yd), (x.grad, xd.grad), (w.grad, wd.grad), (b.grad, bd.grad)):
        assert float((got.double() - want).abs().max()) <= TOL * float(want.abs().max()) + 1e-6


def test_attention_fp32_split_matches_fp64():
    from refign_amd import split32
    q, k, v = _r((2, 2, 130, 32), 9).requires_grad_(True), _r((2, 2, 70, 32), 10).requires_grad_(True), \
        _r((2, 2, 70, 32), 11).requires_grad_(True)
    o = split32.attention(q, k, v, 0.17)
    go = _r(tuple(o.shape), 12)
    o.backward(go)
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    od = torch.softmax(qd @ kd.transpose(-1, -2) * 0.17, -1) @ vd
    od.backward(go.double())
    for got, want in ((o, od), (q.grad, qd.grad), (k.grad, kd.grad), (v.grad, vd.grad)):
        assert float((got.double() - want).abs().max()) <= 4 * TOL * float(want.abs().max()) + 1e-6


def test_fp32_goldens_stay_off_the_libraries():
    """One fp32 forward + backward of MiT-b0 + DAFormer head + HRDA scale attention on HIP tensors records no dense library
    call (F.linear / torch.mm / F.conv2d / scaled_dot_product_attention): the fp32 parity mode the golden-vector tests run
    in is on the hand-written kernels."""
    from refign_amd import mfma
    from refign_amd.seg import DAFormerHead, MixVisionTransformer
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = MixVisionTransformer("mit_b0", drop_path_rate=0.0).to(dev).train()
    head = DAFormerHead([32, 64, 160, 256], [0, 1, 2, 3], 19, 'multiple_select', dropout_ratio=0.0).to(dev).train()
    mfma.LIBRARY_CALLS.clear()
    x = torch.randn(2, 3, 64, 96, device=dev)
    y = head(net(x))
    y.float().square().mean().backward()
    assert not mfma.LIBRARY_CALLS, mfma.library_summary()
