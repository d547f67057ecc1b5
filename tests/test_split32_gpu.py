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


ATTN = [  # B, heads, Nq, Nkv, head dimension  (ragged query / key tails, one / several query chunks in dK / dV)
    (2, 2, 130, 70, 64), (1, 5, 2040, 510, 64), (3, 1, 517, 33, 64), (1, 8, 510, 510, 64), (2, 1, 4100, 480, 64), (1, 2, 31, 5, 64),
    (2, 2, 130, 70, 32), (2, 1, 1536, 24, 32), (1, 8, 96, 96, 32),
]


@pytest.mark.parametrize("B,h,N,Nkv,D", ATTN)
def test_attention_fp32_kernel_matches_fp64(B, h, N, Nkv, D):
    """csrc/attn32.hip (fp32 matrix pipe, fp32 softmax) against the fp64 formulation of mix_transformer.py:147-160 on the
    tensors as the Linears hand them over -- q (B, N, h 64), kv (B, Nkv, 2 h 64) -- forward and the three gradients; the
    forward twice, bit-identical (no atomics there)."""
    from refign_amd import split32
    C = h * D
    q, kv = _r((B, N, C), 9).requires_grad_(True), _r((B, Nkv, 2 * C), 10).requires_grad_(True)
    scale = 0.125
    o = split32.attention(q, kv, h, scale)
    assert o is not None and tuple(o.shape) == (B, N, C)
    assert torch.equal(o, split32.attention(q.detach(), kv.detach(), h, scale))
    go = _r((B, N, C), 12)
    o.backward(go)
    qd, kvd = (t.detach().double().requires_grad_(True) for t in (q, kv))
    k, v = kvd.view(B, Nkv, 2, h, D).permute(2, 0, 3, 1, 4).unbind(0)
    od = (torch.softmax(qd.view(B, N, h, D).transpose(1, 2) @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(B, N, C)
    od.backward(go.double())
    for name, got, want in (("o", o, od), ("dq", q.grad, qd.grad), ("dkv", kv.grad, kvd.grad)):
        err = float((got.double() - want).abs().max())
        assert err <= 2e-5 * float(want.abs().max()) + 1e-7, (name, err, float(want.abs().max()))


def test_attention_fp32_kernel_strided_operands_and_large_scores():
    """q / kv as channel slices of wider tensors (row stride > heads * 64) and scores of +-60 (the running maximum has to carry
    the softmax): still the fp64 result."""
    from refign_amd import split32
    B, h, N, Nkv, C = 2, 2, 300, 96, 128
    qw, kvw = _r((B, N, C + 64), 21, 3.0), _r((B, Nkv, 2 * C + 32), 22, 3.0)
    q, kv = qw[:, :, 64:], kvw[:, :, :2 * C]
    o = split32.attention(q, kv, h, 0.5)
    qd, kvd = q.double(), kv.double()
    k, v = kvd.reshape(B, Nkv, 2, h, 64).permute(2, 0, 3, 1, 4).unbind(0)
    od = (torch.softmax(qd.reshape(B, N, h, 64).transpose(1, 2) @ k.transpose(-1, -2) * 0.5, -1) @ v).transpose(1, 2).reshape(B, N, C)
    assert float((o.double() - od).abs().max()) <= 2e-5 * float(od.abs().max())


@pytest.mark.parametrize("rows,K,pad", [(300, 64, 64), (77, 100, 64), (1000, 86, 8), (5, 3, 8), (4096, 320, 64)])
def test_split3_kernel_equals_the_two_term_split(rows, K, pad):
    """csrc/split3.hip: hi = bf16(x), lo = bf16(x - hi) as three terms in one launch == the torch formulation, bit for bit, side by
    side and stacked, from a row-strided view too; pad columns zero."""
    from refign_amd import split32
    wide = _r((rows, K + 12), 30, 7.0)
    for x in (wide[:, :K].contiguous(), wide[:, 4:K + 4]):
        hi, lo = split32.split2(x)
        Kp = -(-K // pad) * pad
        for order, terms in (("hhl", (hi, hi, lo)), ("hlh", (hi, lo, hi))):
            side = split32.split3(x, order, Kp)
            stack = split32.split3(x, order, Kp, stack=True)
            assert tuple(side.shape) == (rows, 3 * Kp) and tuple(stack.shape) == (3 * rows, Kp)
            for i, t in enumerate(terms):
                assert torch.equal(side[:, i * Kp:i * Kp + K], t) and torch.equal(stack[i * rows:(i + 1) * rows, :K], t)
                assert not side[:, i * Kp + K:(i + 1) * Kp].any() and not stack[i * rows:(i + 1) * rows, K:].any()


@pytest.mark.parametrize("B,H,W,chans", [(2, 19, 37, (81, 2, 2, 1)), (1, 16, 64, (6, 32, 1, 2)), (1, 5, 7, (3,)), (2, 33, 65, (81, 2, 1))])
def test_conv2d_on_parts_equals_conv2d_on_the_concatenation(B, H, W, chans):
    """split32.conv2d_parts (csrc/split3.hip split3_cat_kernel): the decoders' cat(correlation, flow, [feature,] log-variance)
    handed over as parts in MIXED layouts (NCHW, channels-last, a channel slice of a wider channels-last tensor) -- the split
    operand is bit-identical to the one made from torch.cat, so the convolution is too."""
    from refign_amd import split32
    parts = []
    for i, c in enumerate(chans):
        t = _r((B, c, H, W), 40 + i, 50.0)
        if i % 3 == 1:
            t = t.contiguous(memory_format=torch.channels_last)
        elif i % 3 == 2:
            wide = _r((B, H, W, c + 5), 50 + i, 50.0)
            t = wide[..., 2:2 + c].permute(0, 3, 1, 2)
        parts.append(t)
    C, N = sum(chans), 24
    w, b = _r((N, C, 3, 3), 60, (9 * C) ** -0.5), _r((N,), 61)
    with torch.no_grad():
        x3, Cp, Cc = split32.cat_split(parts)
        want3, _ = split32._nhwc3(torch.cat(parts, 1), "hhl")
        assert Cc == C and torch.equal(x3, want3)
        y = split32.conv2d_parts(parts, w, b, 1, 1, 1, act=3)
        want = split32.conv2d(torch.cat(parts, 1), w, b, 1, 1, 1, act=3)
    assert y is not None and torch.equal(y, want)


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
