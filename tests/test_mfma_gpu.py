"""GPU numerics of the hand-written matrix-core kernels (csrc/mfma_gemm.hip, csrc/attn.hip) through the C ABI, each
against a plain PyTorch fp32 reference of the same op fed with the SAME 16-bit-rounded inputs (so the only differences
are fp32 accumulation order and the final 16-bit rounding).  Tolerances are written per test.

Reference semantics: nn.Linear / Mlp / Attention of models/backbones/mix_transformer.py:96-103,137-164."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

EPS = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}


def _rand(shape, dev, dtype, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(300, 64, 64), (129, 256, 64), (1000, 320, 320), (513, 640, 320),
                                   (4111, 1280, 320), (2040, 512, 2048), (77, 8, 128), (260, 24, 64)])
def test_gemm_nt_matches_fp32_reference(dev, dtype, M, N, K):
    """y = x W^T + b: asymmetric random operands (a transposed / swapped tile cannot pass), ragged M, N not a multiple
    of the 64 / 128 tile.  Error bound: one 16-bit rounding of the result + fp32 accumulation noise."""
    from refign_amd.mfma import gemm_nt
    x = _rand((M, K), dev, dtype, 1)
    w = _rand((N, K), dev, dtype, 2, K ** -0.5)
    b = _rand((N,), dev, dtype, 3)
    want = x.float() @ w.float().t() + b.float()
    got = gemm_nt(x, w, b)
    assert got is not None and got.dtype == dtype and got.shape == (M, N)
    tol = 2 * EPS[dtype] * float(want.abs().max())
    assert float((got.float() - want).abs().max()) <= tol
    # no bias, strided operands (views of wider buffers)
    xb = torch.zeros((M, K + 64), dtype=dtype, device=dev)
    xb[:, :K] = x
    got2 = gemm_nt(xb[:, :K], w)
    want2 = x.float() @ w.float().t()
    assert float((got2.float() - want2).abs().max()) <= 2 * EPS[dtype] * float(want2.abs().max())


@pytest.mark.parametrize("act", [1, 3])
def test_gemm_nt_epilogues(dev, act):
    """Fused epilogue: act(x W^T + b), and the stochastic-depth residual res + mask[b] * (x W^T + b)."""
    from refign_amd.mfma import gemm_nt
    dtype = torch.bfloat16
    B, T, N, K = 3, 170, 320, 128
    x = _rand((B * T, K), dev, dtype, 4)
    w = _rand((N, K), dev, dtype, 5, K ** -0.5)
    b = _rand((N,), dev, dtype, 6)
    z = x.float() @ w.float().t() + b.float()
    want = torch.relu(z) if act == 1 else torch.nn.functional.leaky_relu(z, 0.1)
    got = gemm_nt(x, w, b, act=act)
    assert float((got.float() - want).abs().max()) <= 2 * EPS[dtype] * float(want.abs().max())
    res = _rand((B * T, N), dev, dtype, 7)
    mask = torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9], device=dev)
    got = gemm_nt(x, w, b, res=res, rowscale=mask, rows_per_sample=T)
    want = res.float() + mask.repeat_interleave(T)[:, None] * z
    assert float((got.float() - want).abs().max()) <= 2 * EPS[dtype] * float(want.abs().max())
    assert torch.equal(got[:T], res[:T])                       # dropped sample: the residual passes through exactly


@pytest.mark.parametrize("M,N,K,bias,res,scale", [
    (40001, 320, 320, True, False, False),     # ragged bottom edge (40001 = 208 * 192 + 65), one column tile
    (20400, 640, 320, False, False, False),    # the teacher's kv projection, no bias
    (12100, 1280, 192, True, False, False),    # minimum K (three K-steps carry the previous tile's stores)
    (39000, 320, 1280, True, True, True),      # fc2 with the stochastic-depth residual
    (38500, 320, 320, False, True, False),     # residual without bias / scale
    (38500, 320, 128 + 64, True, False, True), # per-sample scale without residual
    # N % 256 == 0 (not a multiple of 320): 192 x 256 tiles, two store-carrying K-steps
    (40001, 256, 1024, False, False, False),   # decode-head 1 x 1 convolution (ASPP branch), ragged bottom edge
    (39000, 512, 256, True, False, False),     # minimum K of the 256-column tiles
    (20400, 2048, 512, True, False, False),    # stage-4 fc1
    (20400, 512, 2048, True, True, True),      # stage-4 fc2 with the stochastic-depth residual
    (38500, 256, 256, False, True, False),     # residual, minimum K
])
def test_gemm_nt_second_generation_kernel(dev, M, N, K, bias, res, scale):
    """Big-M, N % 320 == 0 problems run the software-pipelined kernel of csrc/gemm2.h (192 x 320 tiles, stores issued under
    the next tile's MFMAs, counted s_waitcnt hand-off).  Checked two ways: against the fp32 reference (bound: one 16-bit
    rounding of the result, two with a residual -- the kernel rounds before the residual add like the first generation), and
    BITWISE against the same product computed in row chunks that are too small for the new kernel (< 200 tiles: they
    run gemm_nt_kernel) -- every output row depends on its own row of x only, both kernels walk k in the same order with
    the same MFMA, so the results must be identical; a stale LDS stage, a lost store or a tile mix-up cannot pass."""
    from refign_amd.mfma import gemm_nt
    dtype = torch.bfloat16
    x = _rand((M, K), dev, dtype, 11)
    w = _rand((N, K), dev, dtype, 12, K ** -0.5)
    b = _rand((N,), dev, dtype, 13) if bias else None
    r = _rand((M, N), dev, dtype, 14) if res else None
    rps = 2040
    rs = (0.5 + torch.rand((M + rps - 1) // rps, device=dev)).float() if scale else None
    if rs is not None:
        rs[1] = 0.0                                                   # a dropped sample
    got = gemm_nt(x, w, b, res=r, rowscale=rs, rows_per_sample=rps if scale else 0)
    assert got is not None
    z = x.float() @ w.float().t()
    if b is not None:
        z = z + b.float()
    want = z
    if rs is not None:
        want = rs.repeat_interleave(rps)[:M, None] * want
    if r is not None:
        want = r.float() + want
    tol = (3 if (res or scale) else 2) * EPS[dtype] * float(want.abs().max())
    assert float((got.float() - want).abs().max()) <= tol
    # row chunks of 190 * 192 / (N / 320) rows at most: below the tile threshold of the new kernel; chunk boundaries are
    # multiples of rows_per_sample so that the scale index restarts correctly
    step = rps * max(1, (150 * 192 // (N // (320 if N % 320 == 0 else 256))) // rps)
    parts = []
    for m0 in range(0, M, step):
        m1 = min(M, m0 + step)
        parts.append(gemm_nt(x[m0:m1], w, b, res=None if r is None else r[m0:m1],
                             rowscale=None if rs is None else rs[m0 // rps:].contiguous(), rows_per_sample=rps if scale else 0))
    chunked = torch.cat(parts)
    diff = (got.float() - chunked.float()).abs()
    if res or scale:
        # the residual add is `r + s * v` in both kernels, contracted to an fma by the compiler in one of them: a handful
        # of last-bit differences are legitimate; anything structural would be O(1) wrong on O(tile) entries
        assert int((diff > 0).sum()) <= M * N // 100000 + 8 and float(diff.max()) <= 2 * EPS[dtype] * float(want.abs().max())
    else:
        assert torch.equal(got, chunked)


CONV_CASES = [  # B, H, W, C, N, k, stride, pad, dil
    (2, 17, 23, 64, 64, 3, 1, 1, 1),        # VGG-style 3x3
    (1, 33, 40, 8, 64, 7, 4, 3, 1),         # MiT patch_embed1 (RGB padded to 8 channels): K = 392 -> padded 448
    (2, 20, 28, 64, 128, 3, 2, 1, 1),       # MiT patch_embed2..4
    (1, 24, 24, 32, 128, 3, 1, 4, 4),       # refinement module, dilation 4 (C = 32: two taps per 16-byte... per k-step)
    (1, 19, 27, 88, 128, 3, 1, 1, 1),       # flow decoder (84 channels padded to 88)
    (2, 16, 30, 320, 320, 2, 2, 0, 1),      # spatial-reduction conv, kernel = stride
    (1, 45, 60, 1024, 256, 3, 1, 1, 1),     # DAFormer bottleneck
    (1, 30, 30, 256, 24, 1, 1, 0, 1),       # 1x1 to 19 classes (padded to 24)
    (3, 9, 11, 16, 8, 3, 1, 2, 2),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W,C,N,k,stride,pad,dil", CONV_CASES)
def test_conv2d_implicit_gemm_matches_fp32_reference(dev, dtype, B, H, W, C, N, k, stride, pad, dil):
    """Implicit-GEMM convolution on channels-last tensors against fp32 F.conv2d on the same 16-bit inputs: zero padding
    at all four borders, strides, dilations, channel counts that are not multiples of the 64-wide k-step (a k-step spans
    several taps, the last one is padded), bias + LeakyReLU epilogue, output written into a channel slice."""
    from refign_amd.mfma import conv2d_nhwc, pack_conv_weight
    import torch.nn.functional as F
    x = _rand((B, C, H, W), dev, dtype, 30)
    w = _rand((N, C, k, k), dev, dtype, 31, (C * k * k) ** -0.5)
    b = _rand((N,), dev, dtype, 32)
    want = F.leaky_relu(F.conv2d(x.float(), w.float(), b.float(), stride, pad, dil), 0.1).permute(0, 2, 3, 1)
    xh = x.permute(0, 2, 3, 1).contiguous()
    wp = pack_conv_weight(w, dtype)
    got = conv2d_nhwc(xh, wp, b, k, k, stride, pad, dil, act=3)
    assert got is not None and got.shape == want.shape
    assert float((got.float() - want).abs().max()) <= 2 * EPS[dtype] * float(want.abs().max()) + 1e-3
    wide = torch.zeros(want.shape[:3] + (N + 16,), dtype=dtype, device=dev)
    assert conv2d_nhwc(xh, wp, b, k, k, stride, pad, dil, act=3, out=wide[..., 8:8 + N]) is not None
    assert torch.equal(wide[..., 8:8 + N], got) and float(wide[..., :8].abs().max()) == 0.0


GRAD_CASES = CONV_CASES + [  # true channel counts of the student's convolutions (padded inside the autograd function)
    (2, 36, 44, 3, 64, 7, 4, 3, 1),          # MiT patch_embed1 on RGB
    (2, 27, 30, 256, 19, 1, 1, 0, 1),        # 19-class 1x1
    (2, 21, 26, 320, 512, 3, 2, 1, 1),       # MiT patch_embed4
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W,C,N,k,stride,pad,dil", GRAD_CASES)
def test_conv2d_autograd_on_mfma_kernels_matches_fp32_autograd(dev, dtype, B, H, W, C, N, k, stride, pad, dil):
    """conv._ConvMfmaFn (forward = implicit GEMM, data gradient = the same kernel in transposed-gather mode, weight /
    bias gradient = split-T kernel with gathered im2col rows) against fp32 F.conv2d autograd on the same 16-bit operands
    (daformer.py:65-126, mix_transformer.py:210-242).  16-bit rounding of the result is the only difference: output and
    input gradient are rounded to the 16-bit type, the weight gradient is fp32 (sum over up to B*OH*OW products)."""
    import torch.nn.functional as F
    from refign_amd.conv import conv2d_mfma_grad
    if stride & (stride - 1):
        pytest.skip("data gradient needs a power-of-two stride")
    x = _rand((B, C, H, W), dev, dtype, 40).requires_grad_(True)
    w = _rand((N, C, k, k), dev, torch.float32, 41, (C * k * k) ** -0.5).requires_grad_(True)
    b = _rand((N,), dev, torch.float32, 42).requires_grad_(True)
    w16, b16 = w.detach().to(dtype).float().requires_grad_(True), b.detach().to(dtype).float().requires_grad_(True)
    x32 = x.detach().float().requires_grad_(True)
    want = F.conv2d(x32, w16, b16, stride, pad, dil)
    gy = _rand(tuple(want.shape), dev, dtype, 43)
    want.backward(gy.float())
    got = conv2d_mfma_grad(x, w, b, stride, pad, dil, dtype)
    assert got is not None and got.shape == want.shape
    got.backward(gy)
    e = EPS[dtype]
    assert float((got.float() - want).abs().max()) <= 2 * e * float(want.abs().max()) + 1e-3
    assert float((x.grad.float() - x32.grad).abs().max()) <= 2 * e * float(x32.grad.abs().max()) + 1e-3
    T = want.numel() // N
    tol = 1e-5 * math.sqrt(T / 1000 + 1)
    assert float((w.grad - w16.grad).abs().max()) <= tol * float(w16.grad.abs().max()) + 1e-3
    assert float((b.grad - b16.grad).abs().max()) <= tol * float(b16.grad.abs().max()) + 1e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,N,K,rows", [(8160, 320, 320, None), (2040, 512, 2048, None), (1000, 64, 256, 96),
                                        (4111, 128, 128, 512), (129600, 64, 64, None)])
def test_gemm_tn_matches_fp32_reference(dev, dtype, T, N, K, rows):
    """Weight-gradient GEMM dW = g^T x as per-slab fp32 partials (ragged T: the last slab and the last 32-row stage are
    masked)."""
    from refign_amd.mfma import gemm_tn
    g = _rand((T, N), dev, dtype, 8)
    x = _rand((T, K), dev, dtype, 9)
    part = gemm_tn(g, x, rows)
    assert part is not None and part.dtype == torch.float32 and part.shape[1:] == (N, K)
    want = g.double().t() @ x.double()
    got = part.double().sum(0)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()) * math.sqrt(T / 1000 + 1) + 1e-3


@pytest.mark.parametrize("T,N,K", [(8160, 320, 320), (2040, 512, 2048), (4111, 128, 64)])
def test_gemm_tn_accumulates_into_gradient_views(dev, T, N, K):
    """accumulate mode: every slab is added into an existing fp32 (N, K) buffer with fp32 atomics, and the bias gradient
    (column sums of g) into an existing (N,) buffer -- both on top of what is already there (three backward passes of a
    Refign step accumulate into the same flat gradient buffer)."""
    from refign_amd.mfma import gemm_tn
    dtype = torch.bfloat16
    g = _rand((T, N), dev, dtype, 20)
    x = _rand((T, K), dev, dtype, 21)
    gw0 = torch.randn(N, K, device=dev)
    gb0 = torch.randn(N, device=dev)
    gw, gb = gw0.clone(), gb0.clone()
    assert gemm_tn(g, x, out=gw, bias_out=gb) is gw
    want_w = gw0.double() + g.double().t() @ x.double()
    want_b = gb0.double() + g.double().sum(0)
    assert float((gw.double() - want_w).abs().max()) <= 2e-5 * float(want_w.abs().max()) * math.sqrt(T / 1000 + 1) + 1e-3
    assert float((gb.double() - want_b).abs().max()) <= 2e-5 * float(want_b.abs().max()) * math.sqrt(T / 1000 + 1) + 1e-3


@pytest.mark.parametrize("scaled", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_grouped_weight_gradients_match_fp64(dev, dtype, scaled):
    """Round 5: the weight gradients of a MiT block's Linear layers (stage-3 shapes at the student's 8 160 tokens, plus a ragged
    and a 128-multiple one) queued inside mfma.deferred_wgrads() and launched as ONE group (rfn_gemm_tn_grouped: 64 x 64 tiles,
    fp32 atomics on top of what the gradient views hold, bias column sums; with and without the per-sample stochastic-depth
    scale) against fp64.  Eleven problems: two launches of at most eight."""
    from refign_amd import mfma
    shapes = [(8160, 320, 320), (8160, 640, 320), (8160, 1280, 320), (8160, 320, 1280), (2040, 320, 1280), (4111, 128, 64),
              (2040, 512, 2048), (8160, 64, 64), (8160, 320, 320), (1000, 64, 256), (8160, 320, 320)]
    probs = []
    for i, (T, N, K) in enumerate(shapes):
        g = _rand((T, N), dev, dtype, 30 + i)
        x = _rand((T, K), dev, dtype, 60 + i)
        gw0, gb0 = torch.randn(N, K, device=dev), torch.randn(N, device=dev)
        rps = T // 4 + 1
        rs = (torch.rand(4, device=dev) + 0.5) if scaled else None
        probs.append((g, x, gw0, gb0, gw0.clone(), gb0.clone() if i % 3 else None, rs, rps))
    with mfma.deferred_wgrads():
        for g, x, gw0, gb0, gw, gb, rs, rps in probs:
            assert mfma.defer_gemm_tn(g, x, gw, gb, rs, rps if scaled else 0)
        assert len(mfma._WGRAD_QUEUE) == len(probs)
        assert torch.equal(probs[0][4], probs[0][2]), "nothing may be launched before the flush"
    assert mfma._WGRAD_QUEUE is None
    for g, x, gw0, gb0, gw, gb, rs, rps in probs:
        T = g.shape[0]
        gd = g.double()
        if rs is not None:
            gd = gd * rs.double().repeat_interleave(rps)[:T, None]
        want_w = gw0.double() + gd.t() @ x.double()
        tol = 2e-5 * math.sqrt(T / 1000 + 1)
        assert float((gw.double() - want_w).abs().max()) <= tol * float(want_w.abs().max()) + 1e-3
        if gb is not None:
            want_b = gb0.double() + gd.sum(0)
            assert float((gb.double() - want_b).abs().max()) <= tol * float(want_b.abs().max()) + 1e-3


def test_deferred_weight_gradients_equal_immediate_ones_through_autograd(dev, monkeypatch):
    """A MiT block's backward with its Linear weight gradients queued and launched at the block's mark == the same backward with
    every weight gradient launched where autograd reaches it (gradient views of a flat buffer, bf16): input and parameter gradients
    agree to the rounding of the atomics' order (attention's dK / dV and the weight-gradient slabs are summed with fp32 atomics, so
    two runs of EITHER form differ in the last bits)."""
    from refign_amd import mfma, seg
    from refign_amd.params import mark_grad_sink
    torch.manual_seed(0)
    blk = seg.Block(320, 5, sr_ratio=2).to(dev).train()
    flat = {}
    for n, p in blk.named_parameters():
        p.grad = torch.zeros_like(p)
        mark_grad_sink(p)
    x0 = torch.randn(2, 34 * 60, 320, device=dev)
    res = {}
    for mode in (True, False):
        for p in blk.parameters():
            p.grad.zero_()
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(x, 34, 60)
        loss = (y.float() ** 2).mean()
        if mode:
            with mfma.deferred_wgrads():
                loss.backward()
        else:
            loss.backward()
        res[mode] = (x.grad.clone(), {n: p.grad.clone() for n, p in blk.named_parameters()})
    assert float((res[True][0] - res[False][0]).abs().max()) <= 2e-2 * float(res[False][0].abs().max())
    for n, gdef in res[True][1].items():
        gimm = res[False][1][n]
        assert float(gimm.abs().max()) > 0, n
        assert float((gdef - gimm).abs().max()) <= 2e-2 * float(gimm.abs().max()) + 1e-7, n


def _ref_attention(q, kv, heads, scale):
    B, N, C = q.shape
    d = C // heads
    qh = q.view(B, N, heads, d).transpose(1, 2)
    k, v = kv.view(B, -1, 2, heads, d).permute(2, 0, 3, 1, 4).unbind(0)
    a = ((qh @ k.transpose(-2, -1)) * scale).softmax(-1)
    return (a @ v).transpose(1, 2).reshape(B, N, C)


ATTN_CASES = [  # B, heads, N, Nkv
    (2, 1, 510, 510), (2, 2, 2040, 510), (1, 5, 2040, 510), (2, 8, 510, 510), (1, 1, 32400, 480), (1, 8, 2040, 2040),
    (2, 2, 333, 70), (1, 1, 31, 33), (1, 2, 8160, 510),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,heads,N,Nkv", ATTN_CASES)
def test_attention_forward_backward_vs_fp32_reference(dev, dtype, B, heads, N, Nkv):
    """softmax(scale q k^T) v and its three gradients against fp32 autograd of the textbook formulation on the same
    16-bit inputs: ragged query / key counts (masking of the last key block, partial query tiles), 1-8 heads.
    Bounds: output 2 ulp(16-bit) of its range; gradients 2 % of their range (P and dS are rounded to 16 bit before the
    second GEMM of each chain, as in every flash-attention backward)."""
    from refign_amd.mfma import attention
    C = heads * 64
    scale = 64 ** -0.5
    q = _rand((B, N, C), dev, dtype, 10, 1.5).requires_grad_(True)
    kv = _rand((B, Nkv, 2 * C), dev, dtype, 11, 1.5).requires_grad_(True)
    go = _rand((B, N, C), dev, dtype, 12)
    o = attention(q, kv, heads, scale)
    assert o is not None and o.dtype == dtype
    o.backward(go)
    qf, kvf = q.detach().float().requires_grad_(True), kv.detach().float().requires_grad_(True)
    of = _ref_attention(qf, kvf, heads, scale)
    of.backward(go.float())
    e = EPS[dtype]
    assert float((o.float() - of).abs().max()) <= 4 * e * float(of.abs().max()) + 1e-3
    for name, got, want in (("dq", q.grad, qf.grad), ("dkv", kv.grad, kvf.grad)):
        err = float((got.float() - want).abs().max())
        assert err <= 0.02 * float(want.abs().max()) + 1e-4, (name, err, float(want.abs().max()))


def test_attention_spiked_scores_and_no_grad(dev):
    """One key dominating one query late in the key stream (forces the running-max rescale of the online softmax in a
    late stage), and the gradient-free path (no backward packs)."""
    from refign_amd.mfma import attention
    B, heads, N, Nkv, C = 1, 2, 200, 300, 128
    q = _rand((B, N, C), dev, torch.bfloat16, 13)
    kv = _rand((B, Nkv, 2 * C), dev, torch.bfloat16, 14)
    kv[0, 290, :64] = 8 * q[0, 7, :64]                         # key 290 of head 0 ~ 8 |q_7|^2
    with torch.no_grad():
        o = attention(q, kv, heads, 0.125)
    want = _ref_attention(q.float(), kv.float(), heads, 0.125)
    assert float((o.float() - want).abs().max()) <= 4 * EPS[torch.bfloat16] * float(want.abs().max()) + 1e-3


def test_linear_module_uses_mfma_kernels(dev):
    """refign_amd.linear.Linear under bf16 autocast: forward, input gradient and the parameter gradients (accumulated
    into .grad) against fp32 nn.functional.linear; tokens (B, N, C) like MiT's."""
    from refign_amd.linear import Linear
    torch.manual_seed(0)
    lin = Linear(320, 1280).to(dev)
    x = torch.randn(4, 2040, 320, device=dev, requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = lin(x)
    assert y.dtype == torch.bfloat16
    gy = torch.randn_like(y)
    y.backward(gy)
    xr = x.detach().clone().requires_grad_(True)
    w, b = lin.weight.detach().clone().requires_grad_(True), lin.bias.detach().clone().requires_grad_(True)
    yr = torch.nn.functional.linear(xr, w, b)
    yr.backward(gy.float())
    assert float((y.float() - yr).abs().max()) <= 0.02 * float(yr.abs().max())
    assert float((x.grad - xr.grad).abs().max()) <= 0.02 * float(xr.grad.abs().max())
    assert float((lin.weight.grad - w.grad).abs().max()) <= 0.02 * float(w.grad.abs().max())
    assert float((lin.bias.grad - b.grad).abs().max()) <= 0.02 * float(b.grad.abs().max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,C,H,W,relu", [(2, 256, 17, 23, True), (4, 64, 9, 5, False), (1, 1024, 12, 20, True),
                                          (2, 96, 33, 31, 3), (3, 32, 8, 8, 3)])
def test_batchnorm_relu_train_kernels(dev, dtype, B, C, H, W, relu):
    """csrc/bn.hip against fp32 nn.BatchNorm2d(train) [+ ReLU] on the same 16-bit input: output, input gradient, affine
    gradients, running statistics (unbiased variance, momentum 0.1) and num_batches_tracked."""
    from refign_amd.bn import bn_act_train
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm2d(C).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    ref = torch.nn.BatchNorm2d(C).to(dev)
    ref.load_state_dict(bn.state_dict())
    x = (_rand((B, C, H, W), dev, dtype, 40, 2.0) + 0.7).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = _rand((B, C, H, W), dev, dtype, 41)
    y = bn_act_train(x, bn, relu, dtype)
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    yr = ref(xr)
    yr = torch.nn.functional.leaky_relu(yr, 0.1) if relu == 3 else (torch.relu(yr) if relu else yr)   # 3 = LeakyReLU(0.1)
    yr.backward(g.float())
    e = EPS[dtype]
    assert float((y.float() - yr).abs().max()) <= 4 * e * float(yr.abs().max()) + 1e-3
    assert float((x.grad.float() - xr.grad).abs().max()) <= 0.02 * float(xr.grad.abs().max()) + 1e-4
    assert float((bn.weight.grad - ref.weight.grad).abs().max()) <= 0.02 * float(ref.weight.grad.abs().max()) + 1e-3
    assert float((bn.bias.grad - ref.bias.grad).abs().max()) <= 0.02 * float(ref.bias.grad.abs().max()) + 1e-3
    assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-4)
    assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-4, atol=1e-5)
    assert int(bn.num_batches_tracked) == 1


def test_decode_heads_bf16_kernel_path_matches_fp32_path(dev, monkeypatch):
    """DAFormerHead and SegFormerHead in TRAIN mode (batch-statistics BatchNorm) under bf16 autocast -- 1x1 convolutions
    on the MFMA Linear kernels, depthwise stencils, fused BatchNorm + ReLU kernels -- against the same modules in fp32 on
    the library path.  bf16 through eight normalised layers is noisy by itself, so the yardstick is the LIBRARY's bf16
    path on the same modules: the kernels' deviation from fp32 (logits, input-feature gradients, parameter gradients)
    may not exceed 2x the library-bf16 deviation (+ 2 % of range)."""
    from fill import closed_form_fill
    from refign_amd import mfma
    from refign_amd.seg import DAFormerHead, SegFormerHead
    dims = [64, 128, 320, 512]
    feats = [_rand((2, c, 32 // s, 48 // s), dev, torch.float32, 50 + i) for i, (c, s) in enumerate(zip(dims, (1, 2, 4, 8)))]

    def run(cls, mode):
        m = closed_form_fill(cls(dims, [0, 1, 2, 3], 19, 'multiple_select', dropout_ratio=0.0)).to(dev).train()
        f = [t.clone().requires_grad_(True) for t in feats]
        monkeypatch.setenv("RFN_BN_KERNEL", "0" if mode == "lib16" else "1")
        monkeypatch.setattr(mfma, "ENABLED", mode != "lib16")
        if mode == "fp32":
            y = m(f)
        else:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(f)
        go = _rand(tuple(y.shape), dev, torch.float32, 60)
        y.float().backward(go)
        out = {"logits": y.detach().float()}
        out.update({f"dfeat{i}": t.grad for i, t in enumerate(f)})
        out.update({"d" + n: p.grad for n, p in m.named_parameters()})
        out.update({"buf/" + n: b.detach().float().clone() for n, b in m.named_buffers()})
        return out

    for cls in (DAFormerHead, SegFormerHead):
        ref, lib, ker = run(cls, "fp32"), run(cls, "lib16"), run(cls, "kernels")
        for k in ref:
            rng = float(ref[k].abs().max()) + 1e-6
            e_lib = float((lib[k].float() - ref[k]).abs().max()) / rng
            e_ker = float((ker[k].float() - ref[k]).abs().max()) / rng
            assert e_ker <= 2.0 * e_lib + 0.02, (cls.__name__, k, e_ker, e_lib)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,K,N,scaled", [(4, 300, 64, 256, True), (3, 77, 320, 320, True), (2, 510, 128, 64, False)])
def test_linear_fused_residual_and_drop_path_under_autograd(dev, dtype, B, T, K, N, scaled, monkeypatch):
    """linear.Linear(x, res=, rowscale=) WITH gradients: y = res + rowscale[b] (x W^T + b) in the GEMM epilogue; backward
    through the kernels that take the scale as an argument (input-gradient GEMM epilogue, weight-gradient operand
    staging, bias gradient) against the fp32 composition: output, d/dx, d/dres, dW, db (16-bit rounding tolerances)."""
    from refign_amd import linear
    from refign_amd.linear import Linear
    from refign_amd.trainer import FlatGradBuffer
    monkeypatch.setattr(linear, "_FUSED_RESIDUAL", True)              # opt-in path (RFN_FUSED_RESIDUAL=1)
    torch.manual_seed(1)
    lin = Linear(K, N).to(dev)
    buf = FlatGradBuffer(list(lin.parameters()))                      # gradient sinks, as in the trainer
    x = _rand((B, T, K), dev, dtype, 50).requires_grad_(True)
    res = _rand((B, T, N), dev, dtype, 51).requires_grad_(True)
    rs = torch.tensor([0.0, 1.25, 1.25, 0.0][:B], device=dev) if scaled else None
    gy = _rand((B, T, N), dev, dtype, 52)
    with torch.autocast("cuda", dtype=dtype):
        y = lin(x, res=res, rowscale=rs)
    assert y.dtype == dtype
    y.backward(gy)
    xr, rr = x.detach().float().requires_grad_(True), res.detach().float().requires_grad_(True)
    w, b = lin.weight.detach().clone().requires_grad_(True), lin.bias.detach().clone().requires_grad_(True)
    br = torch.nn.functional.linear(xr, w, b)
    yr = rr + (br if rs is None else br * rs.view(-1, 1, 1))
    yr.backward(gy.float())
    tol = 0.02
    assert float((y.float() - yr).abs().max()) <= tol * float(yr.abs().max())
    assert float((x.grad.float() - xr.grad).abs().max()) <= tol * float(xr.grad.abs().max()) + 1e-6
    assert float((res.grad.float() - rr.grad).abs().max()) <= tol * float(rr.grad.abs().max())
    assert float((lin.weight.grad - w.grad).abs().max()) <= tol * float(w.grad.abs().max()) + 1e-6
    assert float((lin.bias.grad - b.grad).abs().max()) <= tol * float(b.grad.abs().max()) + 1e-6
    if scaled:                                                        # dropped samples contribute nothing at all
        assert float(x.grad[0].abs().max()) == 0.0
    del buf


@pytest.mark.parametrize("C,H,W", [(64, 12, 20), (320, 9, 14)])
@pytest.mark.parametrize("fused_res", [False, True])
def test_mix_ffn_gelu_backward_in_the_fc2_dgrad_epilogue(dev, C, H, W, fused_res, monkeypatch):
    """Mix-FFN (mix_transformer.py:79-103) under bf16 autocast with gelu' applied in the epilogue of fc2's input-gradient
    GEMM (rfn_gemm_nt act = 4; erf by Abramowitz-Stegun 7.1.26) against the same module with the separate gelu_backward
    pass and against an fp32 formulation: output, input gradient, every parameter gradient."""
    import torch.nn.functional as F
    from refign_amd import linear, seg
    from refign_amd.trainer import FlatGradBuffer
    monkeypatch.setattr(linear, "_FUSED_RESIDUAL", fused_res)
    torch.manual_seed(C)
    B = 3

    def run(flag):
        monkeypatch.setenv("RFN_FUSED_GELU_BWD", flag)
        torch.manual_seed(C)
        mlp = seg.Mlp(C, 4 * C).to(dev)
        FlatGradBuffer(list(mlp.parameters()))
        x = _rand((B, H * W, C), dev, torch.bfloat16, 60).requires_grad_(True)
        res = _rand((B, H * W, C), dev, torch.bfloat16, 61).requires_grad_(True)
        rs = torch.tensor([1.2, 0.0, 1.2], device=dev)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = mlp(x, H, W, res=res, rowscale=rs)
        y.backward(_rand(tuple(y.shape), dev, torch.bfloat16, 62))
        return mlp, x, res, y

    m1, x1, r1, y1 = run("1")
    m0, x0, r0, y0 = run("0")
    assert torch.equal(y1, y0)
    for a, b in [(x1.grad, x0.grad), (r1.grad, r0.grad)] + [(p.grad, q.grad) for p, q in zip(m1.parameters(), m0.parameters())]:
        assert float((a.float() - b.float()).abs().max()) <= 3e-2 * float(b.float().abs().max()) + 1e-6
    # fp32 reference of the same function
    xr = x1.detach().float().requires_grad_(True)
    w = {k: v.detach().float().requires_grad_(True) for k, v in m1.named_parameters()}
    h = F.linear(xr, w["fc1.weight"], w["fc1.bias"]).view(B, H, W, 4 * C).permute(0, 3, 1, 2)
    h = F.gelu(F.conv2d(h, w["dwconv.dwconv.weight"], w["dwconv.dwconv.bias"], padding=1, groups=4 * C))
    yr = r1.detach().float() + torch.tensor([1.2, 0.0, 1.2], device=dev).view(B, 1, 1) * \
        F.linear(h.permute(0, 2, 3, 1).reshape(B, H * W, 4 * C), w["fc2.weight"], w["fc2.bias"])
    yr.backward(_rand(tuple(yr.shape), dev, torch.bfloat16, 62).float())
    assert float((x1.grad.float() - xr.grad).abs().max()) <= 4e-2 * float(xr.grad.abs().max())
    for k, p in m1.named_parameters():
        assert float((p.grad - w[k].grad).abs().max()) <= 4e-2 * float(w[k].grad.abs().max()) + 1e-4, k
