"""GPU parity: the HIP kernels (through the C ABI) against the golden vectors captured from the reference and against
the CPU oracle on fresh seeded inputs, plus size-independent properties at the benchmark's full sizes."""
import os

import numpy as np
import pytest
import torch
from conftest import golden, golden_names

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ------------------------------------------------------------------ correlation sampler (a1-a5) -> G1
@pytest.mark.parametrize("name", golden_names("corr_"))
def test_corr_fwd_bwd_golden(dev, name):
    from refign_amd import correlation
    g = golden(name)
    a = [int(v) for v in g["args"]]
    in1, in2, go = T(g["in1"], dev), T(g["in2"], dev), T(g["grad_out"], dev)
    out = correlation.forward(in1, in2, *a)
    tol = dict(rtol=1e-4, atol=1e-4) if g["in1"].dtype == np.float32 else dict(rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], **tol)
    g1, g2 = correlation.backward(in1, in2, go, *a)
    np.testing.assert_allclose(g1.cpu().numpy(), g["grad_in1"], **tol)
    np.testing.assert_allclose(g2.cpu().numpy(), g["grad_in2"], **tol)


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (1, 3, 2, 130), (2, 9, 17, 63), (1, 130, 9, 65), (3, 8, 8, 64)])
def test_corr_hot_vs_oracle_ragged(dev, oracle, shape):
    """ragged / tiny / tile-boundary sizes of the hot parameterisation, autograd API"""
    from refign_amd.correlation import spatial_correlation_sample
    rng = np.random.default_rng(sum(shape))
    a = rng.standard_normal(shape).astype(np.float32)
    b = rng.standard_normal(shape).astype(np.float32)
    ta, tb = T(a, dev).requires_grad_(), T(b, dev).requires_grad_()
    out = spatial_correlation_sample(ta, tb, patch_size=9)
    want = oracle.corr_forward(a, b, patch_size=9)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    go = rng.standard_normal(want.shape).astype(np.float32)
    out.backward(T(go, dev))
    w1, w2 = oracle.corr_backward(a, b, go, patch_size=9)
    np.testing.assert_allclose(ta.grad.cpu().numpy(), w1, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(tb.grad.cpu().numpy(), w2, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(1, 8, 8, 32), (2, 16, 19, 36), (2, 24, 33, 68), (1, 128, 17, 100), (3, 8, 5, 4), (2, 40, 70, 132)])
def test_corr_backward_strip_kernel_vs_oracle(dev, oracle, shape):
    """Round 5: the hot parameterisation's backward on register strips (corr9_bwd_strip_kernel: W % 4 == 0, C % 8 == 0, both
    gradients in one launch) against the oracle's backward: ragged rows, partial column tiles, maps smaller than the halo, more
    than one chunk of 8 channels; run twice (deterministic: bit-identical)."""
    from refign_amd import correlation
    rng = np.random.default_rng(sum(shape) + 7)
    a = rng.standard_normal(shape).astype(np.float32)
    b = rng.standard_normal(shape).astype(np.float32)
    B, C, H, W = shape
    go = rng.standard_normal((B, 9, 9, H, W)).astype(np.float32)
    args = (1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)
    g1, g2 = correlation.backward(T(a, dev), T(b, dev), T(go, dev), *args)
    w1, w2 = oracle.corr_backward(a, b, go, patch_size=9)
    np.testing.assert_allclose(g1.cpu().numpy(), w1, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(g2.cpu().numpy(), w2, rtol=1e-4, atol=1e-4)
    h1, h2 = correlation.backward(T(a, dev), T(b, dev), T(go, dev), *args)
    assert torch.equal(g1, h1) and torch.equal(g2, h2)


@pytest.mark.parametrize("shape", [(2, 48, 23, 72), (1, 256, 19, 36), (1, 32, 40, 32), (2, 64, 8, 64), (1, 80, 9, 132), (2, 32, 72, 160), (2, 128, 33, 70)])
def test_corr_channel_split_tiles_vs_oracle(dev, oracle, shape):
    """Maps of up to 256 8 x 32 tiles (at most one workgroup per CU) with C % 16 == 0 run the pipelined kernel with the workgroup's
    two wave groups on the two halves of the channels (KSPLIT: sums joined through the LDS in front of the epilogue): raw volume
    and fused layer against the oracle, ragged rows / columns, 16 ... 128 channels per group (the last shape is large enough for
    the fused layer to stay off the cross-workgroup split of tiny maps)."""
    from refign_amd.correlation import local_correlation_layer, spatial_correlation_sample
    rng = np.random.default_rng(sum(shape) + 1)
    a = oracle.l2_normalize(np.maximum(rng.standard_normal(shape), 0).astype(np.float32) + 1e-3)
    b = oracle.l2_normalize(np.maximum(rng.standard_normal(shape), 0).astype(np.float32) + 1e-3)
    out = spatial_correlation_sample(T(a, dev), T(b, dev), patch_size=9)
    np.testing.assert_allclose(out.cpu().numpy(), oracle.corr_forward(a, b, patch_size=9), rtol=1e-4, atol=1e-5)
    fused = local_correlation_layer(T(b, dev), T(a, dev))
    np.testing.assert_allclose(fused.cpu().numpy(), oracle.local_correlation_layer(b, a), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 32, 135, 240), (2, 48, 130, 236), (2, 32, 129, 228), (2, 64, 135, 240), (2, 96, 131, 240)])
def test_corr_paired_edge_tiles_vs_oracle(dev, oracle, shape):
    """Round 5 (K4 level 2, 2 x C x 135 x 240): a map of MORE than 256 8 x 32 tiles whose width leaves at most half a tile column
    over shares that column band between the two images of a pair (corr9_pipe2_kernel<.., PAIR>: strips 0..3 of an edge tile are
    image n's last columns, strips 4..7 image n + 1's) -- 255 workgroups instead of 272, one per CU, channel split on top.  Raw
    volume and fused layer against the oracle at the production geometry and with ragged rows and a narrower left-over band; 32 / 48
    channels take two wave groups per tile, 64 / 96 four (C % 32 == 0: the sums of groups 1-3 join group 0's through LDS)."""
    from refign_amd.correlation import local_correlation_layer, spatial_correlation_sample
    B, C, H, W = shape
    assert B * -(-W // 32) * -(-H // 8) > 256 and 0 < W % 32 <= 16          # the shapes this test is about
    rng = np.random.default_rng(sum(shape) + 5)
    a = oracle.l2_normalize(np.maximum(rng.standard_normal(shape), 0).astype(np.float32) + 1e-3)
    b = oracle.l2_normalize(np.maximum(rng.standard_normal(shape), 0).astype(np.float32) + 1e-3)
    out = spatial_correlation_sample(T(a, dev), T(b, dev), patch_size=9)
    np.testing.assert_allclose(out.cpu().numpy(), oracle.corr_forward(a, b, patch_size=9), rtol=1e-4, atol=1e-5)
    fused = local_correlation_layer(T(b, dev), T(a, dev))
    np.testing.assert_allclose(fused.cpu().numpy(), oracle.local_correlation_layer(b, a), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", golden_names("corr_"))
def test_corr_half_dispatch(dev, oracle, name):
    """The CUDA reference dispatches half as well (correlation_cuda_kernel.cu:267); `correlation.forward / backward` take
    float16 tensors of EVERY golden parameterisation: products and sums in fp32, one rounding per element -- so the result is
    the fp64 oracle's on the half-rounded operands to within one half rounding of the result's scale."""
    from refign_amd import correlation
    g = golden(name)
    a = [int(v) for v in g["args"]]
    kw = dict(kernel_size=(a[0], a[1]), patch_size=(a[2], a[3]), padding=(a[4], a[5]), dilation=(a[6], a[7]),
              dilation_patch=(a[8], a[9]), stride=(a[10], a[11]))
    h1, h2, hg = (T(g[k].astype(np.float16), dev) for k in ("in1", "in2", "grad_out"))
    d1, d2, dg = (t.cpu().numpy().astype(np.float64) for t in (h1, h2, hg))
    out = correlation.forward(h1, h2, *a)
    assert out.dtype == torch.float16
    want = oracle.corr_forward(d1, d2, **kw)
    eps = 2.0 ** -10
    assert np.abs(out.cpu().numpy().astype(np.float64) - want).max() <= eps * max(np.abs(want).max(), 1e-3)
    g1, g2 = correlation.backward(h1, h2, hg, *a)
    w1, w2 = oracle.corr_backward(d1, d2, dg, **kw)
    for got, ref in ((g1, w1), (g2, w2)):
        assert got.dtype == torch.float16
        assert np.abs(got.cpu().numpy().astype(np.float64) - ref).max() <= eps * max(np.abs(ref).max(), 1e-3)


def test_corr_errors(dev):
    from refign_amd import correlation
    a = torch.zeros(1, 2, 4, 4, device=dev)
    with pytest.raises(RuntimeError):
        correlation.forward(a, torch.zeros(1, 2, 4, 5, device=dev), 1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError):
        correlation.forward(a, a, 9, 9, 1, 1, 0, 0, 1, 1, 1, 1, 1, 1)      # kernel larger than the image
    with pytest.raises(RuntimeError):
        correlation.forward(a.bfloat16(), a.bfloat16(), 1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)   # float32 / float64 / float16 only
    with pytest.raises(RuntimeError):
        correlation.forward(a.transpose(2, 3), a, 1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)   # non-contiguous


def test_corr_full_size_properties(dev):
    """K4 level-1 size (C=128, 270x480): linearity in input1, shift structure, and agreement with torch on a sampled
    set of shifts (size-independent properties; the oracle would take seconds per call here)."""
    from refign_amd.correlation import spatial_correlation_sample
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.randn(1, 128, 270, 480, generator=g).to(dev)
    b = torch.randn(1, 128, 270, 480, generator=g).to(dev)
    a2 = torch.randn(1, 128, 270, 480, generator=g).to(dev)
    o1 = spatial_correlation_sample(a, b, patch_size=9)
    o2 = spatial_correlation_sample(a2, b, patch_size=9)
    o12 = spatial_correlation_sample(a + 2 * a2, b, patch_size=9)
    assert torch.allclose(o12, o1 + 2 * o2, rtol=1e-3, atol=2e-3)
    for (ph, pw) in [(0, 0), (4, 4), (8, 8), (2, 7), (6, 1)]:
        dy, dx = ph - 4, pw - 4
        bs = torch.zeros_like(b)
        ys, ye = max(0, -dy), min(270, 270 - dy)
        xs, xe = max(0, -dx), min(480, 480 - dx)
        bs[:, :, ys:ye, xs:xe] = b[:, :, ys + dy:ye + dy, xs + dx:xe + dx]
        want = (a.double() * bs.double()).sum(1)
        assert torch.allclose(o1[:, ph, pw].double(), want, rtol=1e-4, atol=1e-3), (ph, pw)


# ------------------------------------------------------------------ LocalFeatureCorrelationLayer (a6) -> G2
@pytest.mark.parametrize("name", golden_names("localcorr_"))
def test_local_layer_golden(dev, name):
    from refign_amd.correlation import local_correlation_layer
    g = golden(name)
    out = local_correlation_layer(T(g["source"], dev), T(g["target"], dev))
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-4, atol=1e-5)


def test_local_layer_fused_warp_vs_oracle(dev, oracle):
    """warp -> correlate -> relu -> l2norm in one kernel == oracle.warp followed by the oracle layer"""
    from refign_amd.correlation import local_correlation_layer
    rng = np.random.default_rng(11)
    B, C, H, W = 2, 24, 21, 70
    src = oracle.l2_normalize(rng.standard_normal((B, C, H, W)).astype(np.float32))
    trg = oracle.l2_normalize(rng.standard_normal((B, C, H, W)).astype(np.float32))
    flo = (2.5 * rng.standard_normal((B, 2, H, W))).astype(np.float32)
    flo[0, :, :3, :3] = 40.0   # out of range region
    want = oracle.local_correlation_layer(oracle.warp(src, flo), trg)
    for single in (False, True):
        out = local_correlation_layer(T(src, dev), T(trg, dev), flow=T(flo, dev), single_kernel_warp=single)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-3, atol=2e-5)


# ------------------------------------------------------------------ GlobalFeatureCorrelationLayer (a7) -> G3
@pytest.mark.parametrize("name", ["globalcorr_c64_16x16", "globalcorr_c24_5x7_6x4"])
def test_global_layer_golden(dev, name):
    from refign_amd.modules import GlobalFeatureCorrelationLayer
    g = golden(name)
    out = GlobalFeatureCorrelationLayer(cyclic_consistency=True)(T(g["source"], dev), T(g["target"], dev))
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=2e-4, atol=2e-6)


def test_global_layer_level4(dev):
    from fill import hashed_uniform
    from refign_amd.matching import l2_normalize_channels
    from refign_amd.modules import GlobalFeatureCorrelationLayer
    g = golden("globalcorr_c512_level4")
    src = l2_normalize_channels(T(hashed_uniform((2, 512, 16, 16), "g3/src") - 0.5, dev))
    trg = l2_normalize_channels(T(hashed_uniform((2, 512, 16, 16), "g3/trg") - 0.5, dev))
    out = GlobalFeatureCorrelationLayer()(src, trg).cpu().numpy()
    np.testing.assert_allclose(out[:, ::7, ::3, ::5], g["out_sample"], rtol=5e-4, atol=5e-6)
    assert abs(out.astype(np.float64).sum() - g["checksum"]) < 1e-3 * g["abs_checksum"]


# ------------------------------------------------------------------ warp (a12) -> G6
@pytest.mark.parametrize("name", golden_names("warp_"))
def test_warp_golden(dev, name):
    from refign_amd.matching import warp
    g = golden(name)
    out, mask = warp(T(g["x"], dev), T(g["flow"], dev), return_mask=True)
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-5, atol=2e-5)
    np.testing.assert_array_equal(mask.cpu().numpy(), g["mask"])


def test_matching_misc_golden(dev):
    from refign_amd import matching
    g = golden("matching_misc")
    f = matching.unnormalise_and_convert_mapping_to_flow(T(g["mapping"], dev))
    np.testing.assert_allclose(f.cpu().numpy(), g["flow"], atol=1e-5)
    c = matching.estimate_probability_of_confidence_interval_of_mixture_density(T(g["logvar"], dev))
    np.testing.assert_allclose(c.cpu().numpy(), g["conf"], rtol=1e-5, atol=1e-6)


def test_align_tail_vs_unfused(dev, oracle):
    """fused upsample+confidence+warp == torch bilinear upsample -> oracle confidence -> oracle warp"""
    from refign_amd.matching import align_tail
    rng = np.random.default_rng(3)
    B, H, W, h, w = 2, 52, 76, 13, 19
    logits = rng.standard_normal((B, 19, H, W)).astype(np.float32)
    fq = (4 * rng.standard_normal((B, 2, h, w))).astype(np.float32)
    lq = rng.uniform(-4, 4, (B, 1, h, w)).astype(np.float32)
    fu = torch.nn.functional.interpolate(torch.from_numpy(fq), size=(H, W), mode="bilinear", align_corners=False).numpy()
    lu = torch.nn.functional.interpolate(torch.from_numpy(lq), size=(H, W), mode="bilinear", align_corners=False).numpy()
    want_w, want_m = oracle.warp(logits, fu, return_mask=True)
    warped, mask, cert, flow_up = align_tail(T(logits, dev), T(fq, dev), T(lq, dev), return_flow=True)
    np.testing.assert_allclose(flow_up.cpu().numpy(), fu, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(cert.cpu().numpy(), oracle.confidence_from_logvar(lu), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(warped.cpu().numpy(), want_w, rtol=1e-4, atol=2e-4)
    assert (mask.cpu().numpy() != want_m).mean() < 1e-3    # only pixels within 1 ulp of the border may differ


# ------------------------------------------------------------------ refine (a17, a24) -> G8
@pytest.mark.parametrize("name", golden_names("refine_"))
def test_refine_golden(dev, name):
    from refign_amd.refine import refine
    g = golden(name)
    if bool(g["no_mask"]):
        out = refine(T(g["logits_trg"], dev), T(g["logits_ref"], dev), None, None, gamma=float(g["gamma"]))
    else:
        out = refine(T(g["logits_trg"], dev), T(g["logits_ref"], dev), T(g["mask"], dev), T(g["cert"], dev),
                     gamma=float(g["gamma"]), disable_M=bool(g["disable_M"]), disable_P=bool(g["disable_P"]))
    out = out.cpu().numpy()
    np.testing.assert_allclose(out, g["out"], rtol=1e-4, atol=1e-5)     # north star: 1e-3
    # pixel-exact argmax mask wherever the reference's top-2 margin is above float noise
    srt = np.sort(g["out"], axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > 1e-5
    assert np.array_equal(out.argmax(1)[decided], g["pseudo_label"][decided])
    assert decided.mean() > 0.99


def test_refine_full_size_properties(dev):
    """1080x1920: identical trg/ref logits => output == softmax (blend of equal things); invalid mask => softmax"""
    from refign_amd.refine import refine
    g = torch.Generator(device="cpu").manual_seed(9)
    lt = (2 * torch.randn(2, 19, 1080, 1920, generator=g)).to(dev)
    cert = torch.rand(2, 1, 1080, 1920, generator=g).to(dev)
    mask = torch.ones(2, 1080, 1920, dtype=torch.bool, device=dev)
    sm = torch.softmax(lt, dim=1)
    out = refine(lt, lt.clone(), mask, cert)
    assert torch.allclose(out, sm, rtol=1e-4, atol=1e-6)
    lr = lt.flip(1).contiguous()
    out = refine(lt, lr, torch.zeros_like(mask), cert)
    assert torch.allclose(out, sm, rtol=1e-4, atol=1e-6)
    out = refine(lt, lr, mask, cert)
    assert torch.isfinite(out).all() and (out >= 0).all() and (out <= 1.0001).all()


@pytest.mark.parametrize("shape,size", [((2, 3, 1080, 1920), (256, 256)), ((1, 3, 37, 53), (8, 16)), ((1, 2, 256, 256), (256, 256))])
def test_area_resize_matches_interpolate_area(dev, shape, size):
    from refign_amd.matching import area_resize
    x = torch.randn(*shape, device=dev)
    want = torch.nn.functional.interpolate(x, size=size, mode="area")
    assert torch.allclose(area_resize(x, size), want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,C,H,W", [(2, 128, 7, 9), (1, 128, 64, 65), (2, 256, 5, 13), (1, 256, 33, 31), (2, 512, 16, 16),
                                     (1, 512, 3, 5), (2, 96, 9, 11), (1, 3, 4, 4)])
def test_l2_normalize_channels(dev, B, C, H, W):
    """F.normalize(x, p=2, dim=1) (uawarpc.py:101-108): register-resident kernels for the VGG widths, generic kernel
    otherwise; ragged pixel counts; an all-zero pixel stays zero (eps = 1e-12 clamp)."""
    from fill import hashed_uniform
    from refign_amd.matching import l2_normalize_channels
    x = T(hashed_uniform((B, C, H, W), f"l2n/{B}/{C}/{H}/{W}") * 4 - 2, dev)
    x[0, :, 0, 0] = 0
    got = l2_normalize_channels(x)
    want = torch.nn.functional.normalize(x, p=2, dim=1)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-7), float((got - want).abs().max())
    assert float(got[0, :, 0, 0].abs().max()) == 0.0


@pytest.mark.parametrize("B,C,H,W", [(2, 128, 9, 11), (1, 256, 17, 33), (2, 512, 16, 16), (1, 128, 270, 480), (3, 8, 5, 7),
                                     (1, 1024, 3, 45)])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_l2_normalize_channels_from_channels_last_16bit(dev, B, C, H, W, dt):
    """The layout and precision the matcher's convolutions deliver under the AMP recipe: channels-last 16-bit features ->
    NCHW float32 == F.normalize(x.float(), p=2, dim=1) (uawarpc.py:101-108), ragged pixel counts (tiles of 32), an
    all-zero pixel; and the result is what the composite path (cast, NCHW copy, NCHW kernel) gives."""
    from fill import hashed_uniform
    from refign_amd.matching import l2_normalize_channels
    x = (T(hashed_uniform((B, C, H, W), f"l2n16/{B}/{C}/{H}/{W}") * 4 - 2, dev)).to(dt)
    x[0, :, 0, 0] = 0
    xcl = x.contiguous(memory_format=torch.channels_last)
    got = l2_normalize_channels(xcl)
    assert got.dtype == torch.float32 and got.is_contiguous() and tuple(got.shape) == (B, C, H, W)
    want = torch.nn.functional.normalize(x.float(), p=2, dim=1)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-7), float((got - want).abs().max())
    assert float(got[0, :, 0, 0].abs().max()) == 0.0
    composite = l2_normalize_channels(x.float().contiguous())
    assert torch.allclose(got, composite, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("B,C,H,W", [(2, 256, 32, 32), (1, 128, 9, 12), (2, 64, 17, 64), (1, 512, 16, 16), (3, 32, 8, 4),
                                     (2, 512, 32, 32), (2, 256, 64, 64), (1, 192, 20, 40)])
def test_local_correlation_layer_channel_split_matches_one_kernel_path(dev, oracle, B, C, H, W, monkeypatch):
    """Small maps split their channels over several workgroups (csrc/corr.hip launch_corr9_split: partial sums, joined with the
    fused ReLU + L2-norm epilogue by the tile's last workgroup inside the launch when the chunks are >= 64 channels in multiples
    of 32 -- shapes 1, 2, 4, 6, 7, 8 --, by a reduce kernel otherwise): same result as the one-kernel path up to the rounding of
    the chunked channel sum, equal to the CPU oracle within the correlation tolerance, deterministic (also across the 20
    back-to-back calls that re-use the tickets), ragged sizes."""
    from refign_amd import correlation
    rng = np.random.default_rng(B * C + H * W)
    src = rng.standard_normal((B, C, H, W)).astype(np.float32)
    trg = rng.standard_normal((B, C, H, W)).astype(np.float32)
    s = correlation._channel_splits(B, C, H, W)
    assert s > 1, "this size is supposed to take the split path"
    got = correlation.local_correlation_layer(T(src, dev), T(trg, dev))
    ts, tt = T(src, dev), T(trg, dev)
    for _ in range(20):
        again = correlation.local_correlation_layer(ts, tt)
    assert torch.equal(got, again)
    monkeypatch.setenv("RFN_CORR_SPLIT", "0")
    one = correlation.local_correlation_layer(T(src, dev), T(trg, dev))
    assert float((got - one).abs().max()) < 2e-6
    if B * H * W <= 4096:
        want = oracle.local_correlation_layer(src, trg)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------ full-size (K4) properties of the remaining kernels
def test_warp_full_size_properties(dev):
    """feature warp at the K4 level-1 size (2 x 128 x 270 x 480): integer flows are exact shifts with zero fill and the
    strict-inequality mask of matching_utils.py:44-47, linearity in x, and agreement with grid_sample
    (align_corners=True, zeros) on a smooth sub-pixel flow."""
    from refign_amd.matching import warp_nocheck
    g = torch.Generator(device="cpu").manual_seed(21)
    B, C, H, W = 2, 128, 270, 480
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    x2 = torch.randn(B, C, H, W, generator=g).to(dev)
    flo = torch.zeros(B, 2, H, W, device=dev)
    flo[:, 0] = 3.0
    flo[:, 1] = -2.0
    out, mask = warp_nocheck(x, flo, return_mask=True)
    want = torch.zeros_like(x)
    want[:, :, 2:, :W - 3] = x[:, :, :H - 2, 3:]
    # the sample position goes through the reference's normalise / un-normalise round trip in fp32 (matching_utils.py:
    # 33-39), so an integer shift lands within ~1e-4 px of the pixel centre: exact up to that interpolation weight
    assert torch.allclose(out, want, atol=2e-3)
    m = mask.view(B, H, W).bool()
    assert bool(m[:, 3:, 1:W - 4].all()) and not bool(m[:, :2].any()) and not bool(m[:, :, W - 3:].any())
    coarse = 4.0 * torch.randn(B, 2, 9, 15, generator=g)
    smooth = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=False).to(dev)
    o1, o2 = warp_nocheck(x, smooth), warp_nocheck(x2, smooth)
    assert torch.allclose(warp_nocheck(x + 2 * x2, smooth), o1 + 2 * o2, rtol=1e-4, atol=1e-4)
    xx = torch.arange(W, device=dev).view(1, 1, 1, W).expand(B, 1, H, W) + smooth[:, :1]
    yy = torch.arange(H, device=dev).view(1, 1, H, 1).expand(B, 1, H, W) + smooth[:, 1:]
    grid = torch.cat((2 * xx / (W - 1) - 1, 2 * yy / (H - 1) - 1), 1).permute(0, 2, 3, 1)
    ref = torch.nn.functional.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    assert torch.allclose(o1, ref, rtol=1e-3, atol=2e-3)


def test_align_tail_full_size_properties(dev):
    """align tail at 1080 x 1920 (x4 bilinear up-sampling of flow and log-variance, confidence, logits warp + mask in
    one kernel): zero flow returns the logits and an all-valid mask; the confidence equals 1 - exp(-1 / (2 exp(lv))) of
    the up-sampled log-variance; the fused result equals the unfused chain interpolate -> warp."""
    from refign_amd.matching import align_tail, warp_nocheck
    g = torch.Generator(device="cpu").manual_seed(22)
    B, H, W = 2, 1080, 1920
    logits = torch.randn(B, 19, H, W, generator=g).to(dev)
    lv = (2 * torch.randn(B, 1, H // 4, W // 4, generator=g)).to(dev)
    warped, mask, cert = align_tail(logits, torch.zeros(B, 2, H // 4, W // 4, device=dev), lv)
    assert torch.allclose(warped[:, :, 1:-1, 1:-1], logits[:, :, 1:-1, 1:-1], atol=2e-3)   # same round trip as above
    # borders: x = 0 / W-1 map to exactly -1 / +1 and are excluded by the strict inequalities of matching_utils.py:44-47
    assert bool(mask[:, 1:-1, 1:-1].all())
    up = torch.nn.functional.interpolate(lv, size=(H, W), mode="bilinear", align_corners=False)
    assert torch.allclose(cert, 1 - torch.exp(-1 / (2 * torch.exp(up))), rtol=1e-4, atol=1e-5)
    coarse = 6.0 * torch.randn(B, 2, 17, 30, generator=g)
    fq = torch.nn.functional.interpolate(coarse, size=(H // 4, W // 4), mode="bilinear", align_corners=False).to(dev)
    warped, mask, cert, fup = align_tail(logits, fq, lv, return_flow=True)
    want_f = torch.nn.functional.interpolate(fq, size=(H, W), mode="bilinear", align_corners=False)
    assert torch.allclose(fup, want_f, rtol=1e-5, atol=1e-4)
    w2, m2 = warp_nocheck(logits, want_f.contiguous(), return_mask=True)
    assert torch.allclose(warped, w2, rtol=1e-3, atol=2e-3)
    assert float((mask.bool().view(-1) != m2.bool().view(-1)).float().mean()) < 1e-5


@torch.no_grad()
def test_uncertainty9_frontend_full_size_properties(dev):
    """fused fp32-MFMA front end of UncertaintyModule at the K4 level-1 size (2 x 81 x 270 x 480): the output of a pixel
    depends on that pixel's 81 values only -- permuting pixels permutes outputs, and a strided subset of pixels run on
    its own gives the same numbers as inside the full map (covers every workgroup / tail position of the full launch)."""
    from fill import closed_form_fill
    from refign_amd import align as A
    um = closed_form_fill(A.UncertaintyModule(1, search_size=9, feed_in_previous=True),
                          "estimate_uncertainty_components1.").to(dev).eval()
    g = torch.Generator(device="cpu").manual_seed(23)
    corr = (torch.rand(2, 81, 270, 480, generator=g) * 2 - 0.5).to(dev)
    full = um.patch_statistics(corr)
    assert full.shape == (2, 6, 270, 480) and bool(torch.isfinite(full).all())
    sub = um.patch_statistics(corr[:, :, 3::7, 5::11].contiguous())
    assert torch.allclose(sub, full[:, :, 3::7, 5::11], rtol=1e-5, atol=1e-6)
    flipped = um.patch_statistics(corr.flip(3).contiguous())
    assert torch.allclose(flipped, full.flip(3), rtol=1e-5, atol=1e-6)


@torch.no_grad()
def test_upsample_concat_full_size(dev):
    """decode-head fusion front end at the student's K4 size (4 views, stage maps 135x240 / 68x120 / 34x60 / 17x30, 256
    channels each, bf16): equals interpolate(bilinear, align_corners=False) + cat level by level."""
    import torch.nn.functional as F
    from refign_amd.upcat import upsample_concat
    g = torch.Generator(device="cpu").manual_seed(24)
    sizes = [(135, 240), (68, 120), (34, 60), (17, 30)]
    toks = [torch.randn(4, h * w, 256, generator=g).to(dev).bfloat16() for h, w in sizes]
    got = upsample_concat(toks, sizes, sizes[0])
    assert got is not None and got.shape == (4, 1024, 135, 240)
    for i, (t, (h, w)) in enumerate(zip(toks, sizes)):
        m = t.float().transpose(1, 2).reshape(4, 256, h, w)
        want = m if i == 0 else F.interpolate(m, size=sizes[0], mode="bilinear", align_corners=False)
        assert torch.allclose(got[:, 256 * i:256 * (i + 1)].float(), want, rtol=2e-2, atol=2e-2), i


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,scale,minr", [(64, 96, 32, 0.75), (1080 // 4, 1920 // 4, 32, 0.75), (40, 56, 8, 0.5),
                                            (33, 47, 16, 0.3)])
def test_label_majority_kernel_matches_one_hot_pooling(H, W, scale, minr):
    """rfn_label_majority (the mask labels of the ImageNet feature-distance loss) == the reference's one-hot ->
    avg_pool2d -> max formulation (segmentation_model.py:637-668), exactly: random labels with ignore pixels, constant
    blocks (ties between equal counts resolve to the smallest class), partial windows at the bottom / right edge."""
    from refign_amd.uda import DomainAdaptationSegmentationModel as M
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H * W)
    gt = torch.randint(0, 19, (2, 1, H, W), generator=g)
    gt[:, :, : H // 2, : W // 2] = torch.randint(0, 19, (2, 1, 1, 1), generator=g)      # large constant regions
    gt[:, :, ::2, W // 2:] = 3                                                          # exact 50 % ties with noise rows
    gt[torch.rand(2, 1, H, W, generator=g) < 0.1] = 255
    oh, ow = -(-H // scale), -(-W // scale)
    want = M.downscale_label_ratio(gt, scale, minr, 19, out_size=(oh, ow))             # CPU: the tensor formulation
    got = M.downscale_label_ratio(gt.to(dev), scale, minr, 19, out_size=(oh, ow))
    assert got.device.type == "cuda" and tuple(got.shape) == (2, 1, oh, ow)
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 18, 22), (1, 128, 17, 19), (3, 8, 2, 2), (1, 256, 33, 62)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_maxpool2x2_kernel_equals_max_pool2d(B, C, H, W, dtype):
    """csrc/warp.hip rfn_maxpool2x2_nhwc16 (the VGG pyramid's nn.MaxPool2d(2, 2), floor mode) on channels-last 16-bit maps,
    odd extents included: bit-equal to F.max_pool2d."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from refign_amd.align import _maxpool2x2
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(B, C, H, W, generator=g).to("cuda:0").to(dtype).contiguous(memory_format=torch.channels_last)
    m = torch.nn.MaxPool2d(kernel_size=2, stride=2)
    with torch.no_grad():
        got = _maxpool2x2(x, m)
    assert got is not None and torch.equal(got, torch.nn.functional.max_pool2d(x, 2, 2))
    assert _maxpool2x2(x.float(), m) is None and _maxpool2x2(x, torch.nn.MaxPool2d(3, 2)) is None


@pytest.mark.parametrize("B,C,h,w,k,dt", [(2, 32, 5, 7, 7, torch.float16), (1, 16, 9, 4, 3, torch.float32), (3, 32, 4, 6, 5, torch.bfloat16),
                                          (1, 8, 3, 3, 1, torch.float16)])
def test_retile_valid_equals_pad_view_slice(dev, B, C, h, w, k, dt):
    """csrc/warp.hip rfn_retile_copy (the uncertainty head's micro-image chain under autograd, models/modules.py:528-545 as tiles of
    one image): the k x k valid part of every (k + 2)-tile == F.pad + 6-d view + slice + reshape, forward and gradient, bit for bit
    (a gather each way)."""
    import torch.nn.functional as F
    from refign_amd.matching import retile_valid
    g = torch.Generator().manual_seed(B * 100 + C + k)
    y = torch.randn(B, C, (k + 2) * h - 2, (k + 2) * w - 2, generator=g).to(dev).to(dt).contiguous(memory_format=torch.channels_last)
    ya, yb = y.clone().requires_grad_(), y.clone().requires_grad_()
    got = retile_valid(ya, h, w, k)
    assert got is not None and tuple(got.shape) == (B, C, k * h, k * w)
    want = F.pad(yb, (0, 2, 0, 2)).view(B, C, h, k + 2, w, k + 2)[:, :, :, :k, :, :k].reshape(B, C, k * h, k * w)
    assert torch.equal(got, want)
    go = torch.randn(B, C, k * h, k * w, generator=g).to(dev).to(dt)
    got.backward(go)
    want.backward(go)
    assert torch.equal(ya.grad, yb.grad)
    assert retile_valid(y.contiguous(), h, w, k) is None               # NCHW memory: the caller's formulation
    # a channel-padded convolution result: the first C of 64 channels per pixel
    wide = torch.zeros(B, (k + 2) * h - 2, (k + 2) * w - 2, 64, device=dev, dtype=dt)
    wide[..., :C] = y.permute(0, 2, 3, 1)
    got2 = retile_valid(wide[..., :C].permute(0, 3, 1, 2), h, w, k)
    assert got2 is not None and torch.equal(got2, want.detach())
