"""GPU: hand-written depthwise 3x3 (csrc/dwconv.hip) forward / backward-data / backward-weight against torch's
conv2d (fp32 reference of the same op), fp32 and bf16 activations, dilations 1/6, ragged sizes."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,C,dil", [(2, 7, 9, 32, 1), (1, 17, 30, 256, 1), (2, 33, 41, 64, 6), (1, 5, 3, 8, 1),
                                        (1, 40, 64, 1024, 12),
                                        # the XCD-sliced geometry (csrc/dwconv.hip sliced_geom): slices of 20 / 40 vectors
                                        # (idle threads in the block), rows in residue-class order with H % dilation != 0
                                        (2, 9, 13, 1280, 1), (1, 23, 31, 1024, 6), (3, 5, 4, 2048, 18),
                                        # the LDS-tiled kernel (bf16, C % 64 == 0): ragged tile edges, several tiles per phase,
                                        # phases of unequal size, one-pixel-wide phase images
                                        (1, 37, 70, 128, 1), (2, 135, 61, 64, 2), (1, 40, 67, 192, 3), (2, 19, 7, 64, 6)])
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


@pytest.mark.parametrize("B,H,W,C", [(2, 9, 13, 64), (2, 9, 13, 1280), (1, 34, 60, 512), (2, 68, 120, 128), (1, 135, 240, 64)])
def test_fused_dwconv_gelu_matches_conv_then_exact_gelu(dev, B, H, W, C):
    """gelu(DWConv(x)) in one pass (mix_transformer.py:99-101) on bf16 tokens: the activation of the 16-bit path is the
    branch-free erf of csrc/mfma.h (A&S 7.1.26, |error| < 1.5e-7) -- compared with the exact-erf GELU of an fp32
    convolution, to the rounding of the bf16 result; the pre-activation z (with_z) is the plain convolution."""
    from refign_amd.dwconv import dwconv3x3_gelu_tokens
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, H * W, C, generator=g) * 1.5).to(dev).to(torch.bfloat16)
    w = (torch.randn(C, 1, 3, 3, generator=g) * 0.4).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    z_ref = torch.nn.functional.conv2d(x.float().view(B, H, W, C).permute(0, 3, 1, 2), w, b, padding=1, groups=C)
    z_ref = z_ref.permute(0, 2, 3, 1).reshape(B, H * W, C)
    a_ref = torch.nn.functional.gelu(z_ref)
    with torch.no_grad():
        a = dwconv3x3_gelu_tokens(x, w, b, H, W)
    assert a.dtype == torch.bfloat16
    tol = 2.0 ** -8 * a_ref.abs() + 2e-3
    assert bool(((a.float() - a_ref).abs() <= tol).all())
    xg = x.clone().requires_grad_()
    a2, z = dwconv3x3_gelu_tokens(xg, w, b, H, W, with_z=True)
    assert torch.equal(a2, a)
    assert bool(((z.float() - z_ref).abs() <= 2.0 ** -8 * z_ref.abs() + 2e-3).all())


@pytest.mark.parametrize("B,H,W,C,dil", [(2, 20, 28, 1024, 6), (3, 9, 13, 1280, 1), (2, 17, 30, 64, 1), (1, 23, 31, 256, 12)])
def test_dwconv_leaves_the_batchnorm_statistics_of_its_result(dev, B, H, W, C, dil):
    """rfn_dwconv3x3_nhwc_fwd_stats (depthwise 3x3 -> BatchNorm of the ASPP branches, daformer.py:10-62): same result as the
    plain kernel, and the (sum, sum of squares, rows) buffer equals what bn._stats_fwd reads back from that result --
    both geometries of the kernel (first-generation grid, XCD-sliced with idle threads)."""
    from refign_amd import bn as bnk
    from refign_amd.dwconv import dwconv3x3_nhwc
    g = torch.Generator().manual_seed(C + H + dil)
    x = (torch.randn(B, H, W, C, generator=g) + 0.3).to(dev).to(torch.bfloat16)
    w = torch.randn(C, 1, 3, 3, generator=g).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    with torch.no_grad():
        want = dwconv3x3_nhwc(x, w, b, dil)
        sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
        got = dwconv3x3_nhwc(x, w, b, dil, stats=sums)
    assert torch.equal(got, want)
    ref = torch.empty_like(sums)
    bnk._stats_fwd(want, ref)
    assert float(sums[2 * C]) == float(ref[2 * C]) == B * H * W
    scale = ref[C:2 * C].abs().max()
    assert float((sums[:C] - ref[:C]).abs().max()) <= 1e-5 * float(ref[:C].abs().max() + scale.sqrt())
    assert float((sums[C:2 * C] - ref[C:2 * C]).abs().max()) <= 1e-5 * float(scale)


@pytest.mark.parametrize("B,H,W,C,dil,relu", [(2, 20, 28, 1024, 6, True), (3, 9, 13, 1280, 1, True), (2, 17, 30, 64, 18, False)])
def test_gradient_free_dwconv_batchnorm_relu_in_two_input_passes(dev, B, H, W, C, dil, relu):
    """dwconv3x3_bn_act_nhwc (statistics without a store, then convolution + BatchNorm(batch statistics) + ReLU) == the
    chain depthwise kernel -> csrc/bn.hip statistics -> apply, result and running buffers (same rounded values enter the
    statistics; the fp64 sums differ only in summation order)."""
    import copy
    from refign_amd import bn as bnk
    from refign_amd.dwconv import dwconv3x3_bn_act_nhwc, dwconv3x3_nhwc
    g = torch.Generator().manual_seed(C + H + dil)
    x = (torch.randn(B, H, W, C, generator=g) + 0.2).to(dev).to(torch.bfloat16)
    w = torch.randn(C, 1, 3, 3, generator=g).to(dev)
    b = torch.randn(C, generator=g).to(dev)
    bn_a = torch.nn.BatchNorm2d(C).to(dev).train()
    with torch.no_grad():
        bn_a.weight.copy_(torch.rand(C, generator=g).to(dev) + 0.5)
        bn_a.bias.copy_(torch.randn(C, generator=g).to(dev))
    bn_b = copy.deepcopy(bn_a)
    with torch.no_grad():
        got = dwconv3x3_bn_act_nhwc(x, w, b, dil, bn_a, relu)
        conv = dwconv3x3_nhwc(x, w, b, dil)
        want = bnk.bn_act_train(conv.permute(0, 3, 1, 2), bn_b, 1 if relu else 0, torch.bfloat16).permute(0, 2, 3, 1)
    err = (got.float() - want.float()).abs()
    assert float(err.max()) <= 2.0 ** -7 * float(want.float().abs().max())          # at most a rounding step apart ...
    assert float((err > 0).float().mean()) < 1e-3                                    # ... and almost everywhere equal
    assert torch.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-6, atol=1e-7)
    assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-6, atol=1e-7)
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 1


@pytest.mark.parametrize("B,H,W,C,g,relu,bias", [(2, 135, 240, 128, 6, True, True), (3, 37, 53, 64, 2, True, False),
                                                   (2, 20, 28, 1024, 6, False, True), (1, 7, 9, 64, 1, True, True)])
def test_three_dilations_of_one_input_in_two_passes(dev, B, H, W, C, g, relu, bias):
    """dwconv3x3_bn_act_nhwc_tri (dilations g, 2 g, 3 g of one input: ONE statistics pass + ONE convolution + BatchNorm + ReLU
    pass, each workgroup computing the three branches from a phase sub-image held in LDS) == three calls of the two-pass
    single-branch path (dwconv3x3_bn_act_nhwc, itself pinned to depthwise kernel -> bn.hip above): results, running buffers,
    batch counters.  Ragged phase sub-images (135 = 22 x 6 + 3, 37 x 53 with g = 2), images smaller than the largest dilation
    (7 x 9 with dilation 3: every off-centre tap of that branch is padding), the teacher's map size."""
    import copy
    import types
    from refign_amd.dwconv import dwconv3x3_bn_act_nhwc, dwconv3x3_bn_act_nhwc_tri, tri_usable
    gen = torch.Generator().manual_seed(C + H + g)
    x = (torch.randn(B, H, W, C, generator=gen) + 0.2).to(dev).to(torch.bfloat16)
    convs, bns_a = [], []
    for k in range(3):
        d = g * (k + 1)
        conv = torch.nn.Conv2d(C, C, 3, padding=d, dilation=d, groups=C, bias=bias).to(dev)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(C, 1, 3, 3, generator=gen).to(dev))
            if bias:
                conv.bias.copy_(torch.randn(C, generator=gen).to(dev))
        bn = torch.nn.BatchNorm2d(C).to(dev).train()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(C, generator=gen).to(dev) + 0.5)
            bn.bias.copy_(torch.randn(C, generator=gen).to(dev))
        convs.append(conv)
        bns_a.append(bn)
    bns_b = copy.deepcopy(bns_a)
    assert tri_usable(x, convs, bns_a)
    assert not tri_usable(x, [convs[0], convs[0], convs[2]], bns_a)                  # dilations must be g, 2 g, 3 g
    with torch.no_grad():
        got = dwconv3x3_bn_act_nhwc_tri(x, convs, bns_a, relu)
        want = [dwconv3x3_bn_act_nhwc(x, c.weight, c.bias, c.dilation[0], b, relu) for c, b in zip(convs, bns_b)]
    for k in range(3):
        err = (got[k].float() - want[k].float()).abs()
        assert float(err.max()) <= 2.0 ** -7 * float(want[k].float().abs().max()), k   # at most a rounding step apart ...
        assert float((err > 0).float().mean()) < 1e-3, k                               # ... and almost everywhere equal
        assert torch.allclose(bns_a[k].running_mean, bns_b[k].running_mean, rtol=1e-6, atol=1e-7)
        assert torch.allclose(bns_a[k].running_var, bns_b[k].running_var, rtol=1e-6, atol=1e-7)
        assert int(bns_a[k].num_batches_tracked) == 1


@pytest.mark.parametrize("views,H,W,C", [(3, 17, 30, 512), (2, 34, 60, 320), (2, 7, 45, 128), (1, 135, 240, 64), (2, 13, 31, 320),
                                         (1, 6, 30, 64), (1, 1, 1, 128), (2, 68, 120, 128)])
def test_fused_mix_ffn_front_half_equals_the_three_kernels(dev, views, H, W, C):
    """Round 6, csrc/mixffn.hip: gelu(dw3x3(fc1(x))) of a Mix-FFN (mix_transformer.py:99-101) in ONE kernel for the gradient-free
    passes -- all four MiT-B5 stage widths, token maps that are not multiples of the 6 x 30 interior tile (ragged right / bottom
    tiles, maps smaller than one tile), several views (no halo leaks across views) -- against the three-kernel formulation
    (GEMM with bf16 store, depthwise + GELU kernel) on the same modules: the same roundings, so at most one bf16 step apart; and
    against the fp32 torch formulation; twice, bit-identical."""
    from refign_amd import dwconv
    from refign_amd.seg import Mlp
    torch.manual_seed(C + H)
    mlp = Mlp(C, 4 * C).to(dev).eval()
    with torch.no_grad():
        for p in mlp.parameters():
            p.mul_(2.0)
        mlp.dwconv.dwconv.bias.normal_(0, 0.5)
        mlp.fc1.bias.normal_(0, 0.5)
    x = torch.randn(views, H * W, C, device=dev).to(torch.bfloat16)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        a = dwconv.ffn_fc1_dw_gelu(x, mlp.fc1, mlp.dwconv.dwconv, H, W)
        assert a is not None and tuple(a.shape) == (views, H * W, 4 * C)
        assert torch.equal(a, dwconv.ffn_fc1_dw_gelu(x, mlp.fc1, mlp.dwconv.dwconv, H, W))
        h = mlp.fc1(x)
        want = dwconv.dwconv3x3_gelu_tokens(h, mlp.dwconv.dwconv.weight, mlp.dwconv.dwconv.bias, H, W)
        y_fused = mlp(x, H, W)
        dwconv.FUSED_FFN = False
        try:
            y_plain = mlp(x, H, W)
        finally:
            dwconv.FUSED_FFN = True
    d = (a.float() - want.float()).abs()
    scale = float(want.float().abs().max())
    assert float(d.max()) <= 2.0 ** -6 * scale and float(d.mean()) <= 2e-4 * scale, (float(d.max()), float(d.mean()), scale)
    assert float((y_fused.float() - y_plain.float()).abs().max()) <= 2.0 ** -5 * float(y_plain.float().abs().max())
    # fp32 formulation (the reference's ops)
    xf = x.float()
    hf = torch.nn.functional.linear(xf, mlp.fc1.weight, mlp.fc1.bias).to(torch.bfloat16).float()
    hf = hf.transpose(1, 2).reshape(views, 4 * C, H, W)
    ref = torch.nn.functional.gelu(torch.nn.functional.conv2d(hf, mlp.dwconv.dwconv.weight, mlp.dwconv.dwconv.bias, padding=1,
                                                              groups=4 * C)).flatten(2).transpose(1, 2)
    assert float((a.float() - ref).abs().max()) <= 3e-2 * float(ref.abs().max())
