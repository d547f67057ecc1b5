"""Repeat-and-compare checks of the kernels whose chunk / stage hand-offs are COUNTED waits (`s_waitcnt vmcnt(n)` + barrier over
LDS-DMA instructions the compiler does not track): the same launch, many times, with another stream keeping the memory system
busy -- every result must equal the first bit for bit.  A hand-off that lets a wave read a ring slot before its DMA has landed
passes every golden on a quiet device and shows up here as a few launches that differ by the products of a chunk or two
(round 4: the first 4-stage correlation kernel did, at the tail of every tile; tools/micro/corr_race.py is the long form)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _repeat(fn, reps, noise_src):
    ref = fn().clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    bad = 0
    for i in range(reps):
        if i % 2 == 0:
            with torch.cuda.stream(side):
                _ = noise_src.clone()            # ~130-260 MB copy in flight next to the kernel under test
        out = fn()
        if not torch.equal(out, ref):
            bad += 1
    torch.cuda.synchronize()
    return bad


@pytest.mark.parametrize("B,C,H,W,reps", [(2, 128, 270, 480, 150), (2, 256, 135, 240, 150), (2, 128, 128, 128, 150), (1, 64, 71, 52, 60)])
def test_local_correlation_is_repeatable_under_memory_load(B, C, H, W, reps):
    """corr9_pipe2_kernel at the K4 level-1 / level-2 and K2 level-1 sizes (16 x 32 paired and 8 x 32 single tiles) and on a
    ragged map: identical results, launch after launch."""
    from refign_amd import correlation
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    f1 = torch.nn.functional.normalize(torch.relu(torch.randn(B, C, H, W, generator=g)), dim=1).to(dev)
    f2 = torch.nn.functional.normalize(torch.relu(torch.randn(B, C, H, W, generator=g)), dim=1).to(dev)
    noise = torch.empty(48 << 20, device=dev, dtype=torch.float32).normal_()
    assert _repeat(lambda: correlation.local_correlation_layer(f2, f1), reps, noise) == 0


@pytest.mark.parametrize("B,C,H,W,reps", [(2, 256, 32, 32, 300), (2, 512, 32, 32, 200), (2, 256, 64, 64, 200), (1, 128, 9, 12, 100)])
def test_joined_channel_split_is_repeatable_under_memory_load(B, C, H, W, reps):
    """Round 5: tiny maps in ONE launch -- the slices of a tile meet through global memory inside the launch (slab stores, agent-scope
    release fence, ticket; the last workgroup's acquire fence, then plain loads; tickets put back to zero for the next launch).  A
    hand-off that publishes the ticket before the slabs, or a ticket that is not back at zero, shows up as a launch that differs:
    identical results for hundreds of back-to-back launches on one workspace, with a copy stream keeping the memory system busy."""
    from refign_amd import correlation
    dev = torch.device("cuda:0")
    assert correlation._channel_splits(B, C, H, W) > 1 and (C // correlation._channel_splits(B, C, H, W)) % 32 == 0
    g = torch.Generator().manual_seed(3)
    f1 = torch.nn.functional.normalize(torch.relu(torch.randn(B, C, H, W, generator=g)), dim=1).to(dev)
    f2 = torch.nn.functional.normalize(torch.relu(torch.randn(B, C, H, W, generator=g)), dim=1).to(dev)
    noise = torch.empty(48 << 20, device=dev, dtype=torch.float32).normal_()
    assert _repeat(lambda: correlation.local_correlation_layer(f2, f1), reps, noise) == 0


@pytest.mark.parametrize("B,C,H,W,reps", [(2, 128, 270, 480, 30), (6, 128, 65, 68, 100), (1, 64, 33, 72, 100)])
def test_correlation_backward_is_repeatable_under_memory_load(B, C, H, W, reps):
    """Round 5: corr9_bwd_strip_kernel (double-buffered halo tiles, one barrier per chunk of 8 channels, the three wave groups' sums
    exchanged through double-buffered LDS): both gradients identical, launch after launch."""
    from refign_amd import correlation
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    f1 = torch.randn(B, C, H, W, generator=g).to(dev)
    f2 = torch.randn(B, C, H, W, generator=g).to(dev)
    go = torch.randn(B, 9, 9, H, W, generator=g).to(dev)
    noise = torch.empty(48 << 20, device=dev, dtype=torch.float32).normal_()
    args = (1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)
    assert _repeat(lambda: torch.cat([t.flatten() for t in correlation.backward(f1, f2, go, *args)]), reps, noise) == 0


@pytest.mark.parametrize("M,N,K,res", [(81600, 320, 320, True), (81600, 320, 1280, True), (20400, 512, 2048, False),
                                       (8160, 1280, 320, False)])
def test_gemm_nt_is_repeatable_under_memory_load(M, N, K, res):
    """gemm_nt2_kernel (counted hand-off behind dripped stores) and gemm_nt_kernel (ring with counted waits): 60 launches."""
    from refign_amd import mfma
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).to(dev).bfloat16()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).bfloat16()
    b = torch.randn(N, generator=g).to(dev).bfloat16()
    r = torch.randn(M, N, generator=g).to(dev).bfloat16() if res else None
    noise = torch.empty(64 << 20, device=dev, dtype=torch.float32).normal_()
    assert _repeat(lambda: mfma.gemm_nt(x, w, b, res=r), 60, noise) == 0


def test_attention_forward_is_repeatable_under_memory_load():
    """attn_fwd_kernel on the teacher's stage-3 shape (40 views x 5 heads, 2040 queries, 510 keys): 40 launches."""
    from refign_amd import mfma
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    q = torch.randn(8, 2040, 320, generator=g).to(dev).bfloat16()
    kv = torch.randn(8, 510, 640, generator=g).to(dev).bfloat16()
    noise = torch.empty(64 << 20, device=dev, dtype=torch.float32).normal_()

    def fn():
        with torch.no_grad():
            return mfma.attention(q, kv, 5, 0.125)
    assert _repeat(fn, 40, noise) == 0


def test_implicit_gemm_convolution_is_repeatable_under_memory_load():
    """gemm_nt_kernel<GATHER> (3 x 3 convolution of the teacher's fusion layer, 4 views here): DMA gather + zero page, ring."""
    from refign_amd import mfma
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 135, 240, 1024, generator=g).to(dev).bfloat16()
    w = (torch.randn(256, 1024, 3, 3, generator=g) / 96).to(dev)
    wp = mfma.pack_conv_weight(w, torch.bfloat16)
    noise = torch.empty(64 << 20, device=dev, dtype=torch.float32).normal_()
    assert _repeat(lambda: mfma.conv2d_nhwc(x, wp, None, 3, 3, pad=1), 30, noise) == 0


@pytest.mark.parametrize("T,N,K", [(8160, 1280, 320), (8160, 320, 320), (129600, 256, 1024)])
def test_weight_gradient_partials_are_repeatable_under_memory_load(T, N, K):
    """gemm_tn3_kernel, deterministic form (fp32 partials per slab): 40 launches."""
    from refign_amd import mfma
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    gy = torch.randn(T, N, generator=g).to(dev).bfloat16()
    x = torch.randn(T, K, generator=g).to(dev).bfloat16()
    noise = torch.empty(64 << 20, device=dev, dtype=torch.float32).normal_()
    assert _repeat(lambda: mfma.gemm_tn(gy, x), 40, noise) == 0


def test_attention_input_gradients_are_repeatable_under_memory_load():
    """attn_bwd_dq_kernel (deterministic); dK / dV are accumulated with fp32 atomics over query blocks, so THEIR bits depend
    on the order of arrival: compared with a tolerance, the query gradient bit for bit."""
    from refign_amd import mfma
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    q0 = torch.randn(4, 2040, 320, generator=g).to(dev).bfloat16()
    kv0 = torch.randn(4, 510, 640, generator=g).to(dev).bfloat16()
    go = torch.randn(4, 2040, 320, generator=g).to(dev).bfloat16()
    noise = torch.empty(64 << 20, device=dev, dtype=torch.float32).normal_()
    side = torch.cuda.Stream()

    def grads():
        q, kv = q0.clone().requires_grad_(True), kv0.clone().requires_grad_(True)
        out = mfma.attention(q, kv, 5, 0.125)
        out.backward(go)
        return q.grad, kv.grad
    rq, rkv = grads()
    torch.cuda.synchronize()
    for i in range(30):
        if i % 2 == 0:
            with torch.cuda.stream(side):
                _ = noise.clone()
        gq, gkv = grads()
        assert torch.equal(gq, rq), f"query gradient differs in launch {i}"
        torch.testing.assert_close(gkv.float(), rkv.float(), rtol=2e-2, atol=2e-2)
    torch.cuda.synchronize()


@pytest.mark.parametrize("views,H,W,C,reps", [(8, 34, 60, 320, 100), (2, 135, 240, 64, 60)])
def test_fused_mix_ffn_front_half_is_repeatable_under_memory_load(views, H, W, C, reps):
    """Round 6: csrc/mixffn.hip lays its hidden tile over the operand ring after the last chunk and hands chunks over through
    registers and two LDS stages with one barrier each: identical activations, launch after launch."""
    from refign_amd import dwconv
    from refign_amd.seg import Mlp
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    mlp = Mlp(C, 4 * C).to(dev).eval()
    x = torch.randn(views, H * W, C, device=dev).to(torch.bfloat16)
    noise = torch.empty(48 << 20, device=dev, dtype=torch.float32).normal_()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        assert _repeat(lambda: dwconv.ffn_fc1_dw_gelu(x, mlp.fc1, mlp.dwconv.dwconv, H, W), reps, noise) == 0
