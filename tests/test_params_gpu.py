"""GPU: parameter plumbing of the training step (refign_amd/params.py, csrc/reduce.hip): leading-dimension sums with
accumulation, versioned bf16 weight copies, and backward kernels that accumulate straight into the flat gradient
buffer -- each against the stock autograd path."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S,n", [(1, 8), (3, 24), (64, 4096), (65, 64), (1000, 320), (8160, 320), (4099, 1280),
                                 (300, 2048), (129, 2056), (32640, 128), (200, 102400)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_sum_rows(dev, S, n, dt):
    from refign_amd.params import sum_rows
    g = torch.Generator().manual_seed(S * 7 + n)
    x = torch.randn(S, n, generator=g).to(dev).to(dt)
    want = x.double().sum(0)
    got = sum_rows(x)
    assert got.dtype == torch.float32 and got.shape == (n,)
    tol = 2e-5 * (S ** 0.5) + 1e-6
    assert torch.allclose(got.double(), want, rtol=1e-5, atol=tol), float((got.double() - want).abs().max())
    # accumulate into an existing buffer; deterministic (bit-identical when repeated)
    base = torch.randn(n, generator=g).to(dev)
    out = base.clone()
    sum_rows(x, out=out, accumulate=True)
    assert torch.allclose(out.double(), base.double() + want, rtol=1e-5, atol=tol)
    out2 = base.clone()
    sum_rows(x, out=out2, accumulate=True)
    assert torch.equal(out, out2)


def test_sum_rows_rejects_bad_width_through_fallback(dev):
    from refign_amd.params import sum_rows
    x = torch.randn(10, 19, device=dev)                      # n % 8 != 0: library path
    assert torch.allclose(sum_rows(x), x.sum(0), atol=1e-5)


def test_derived_copy_follows_in_place_updates(dev):
    from refign_amd.params import as_dtype, refresh
    p = nn.Parameter(torch.randn(16, 8, device=dev))
    a = as_dtype(p, torch.bfloat16)
    assert a.dtype == torch.bfloat16 and as_dtype(p, torch.bfloat16) is a       # cached
    assert as_dtype(p, torch.float32) is p
    with torch.no_grad():
        p.mul_(2.0)                                                             # what an optimizer step does
    b = as_dtype(p, torch.bfloat16)
    assert b is not a and torch.equal(b, p.detach().to(torch.bfloat16))
    p.data.add_(1.0)                                   # through .data (EMA update): invisible to the version counter,
    c = as_dtype(p, torch.bfloat16)                    # so the updater calls refresh(): same buffer, new contents
    refresh([p])
    assert as_dtype(p, torch.bfloat16) is c and torch.equal(c, p.detach().to(torch.bfloat16))
    p.data = torch.zeros_like(p.data)                                           # re-allocation (load / .to())
    assert torch.equal(as_dtype(p, torch.bfloat16), torch.zeros(16, 8, device=dev, dtype=torch.bfloat16))


class _Net(nn.Module):
    """every module type whose backward accumulates into the flat buffer"""

    def __init__(self):
        super().__init__()
        from refign_amd.conv import Conv2d
        from refign_amd.layernorm import LayerNorm
        from refign_amd.linear import Linear
        self.embed = Conv2d(3, 64, 3, stride=2, padding=1)
        self.norm = LayerNorm(64)
        self.fc1 = Linear(64, 128)
        self.dw = nn.Conv2d(128, 128, 3, 1, 1, groups=128)
        self.fc2 = Linear(128, 64)
        self.sr = Conv2d(64, 64, 2, stride=2)

    def forward(self, img):
        from refign_amd.dwconv import dwconv3x3_tokens
        x = self.embed(img)
        B, C, H, W = x.shape
        t = self.norm(x.flatten(2).transpose(1, 2))
        h = dwconv3x3_tokens(self.fc1(t), self.dw.weight, self.dw.bias, H, W)
        t = t + self.fc2(F.gelu(h))
        return self.sr(t.transpose(1, 2).reshape(B, C, H, W))


@pytest.mark.parametrize("autocast", [False, True])
def test_backward_into_flat_gradient_buffer_equals_autograd(dev, autocast):
    """three backward passes accumulated by the kernels into FlatGradBuffer views == stock autograd accumulation"""
    from refign_amd.trainer import FlatGradBuffer
    torch.manual_seed(3)
    a = _Net().to(dev)
    b = _Net().to(dev)
    b.load_state_dict(a.state_dict())
    flat = FlatGradBuffer(a.parameters())
    assert all(getattr(p, "_rfn_grad_sink", False) for p in a.parameters())
    imgs = [torch.randn(2, 3, 96, 80, device=dev) for _ in range(3)]
    for m in (a, b):
        for img in imgs:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                m(img).float().square().mean().backward()
    assert all(p.grad.data_ptr() >= flat.flat.data_ptr() for p in flat.params)              # still the views
    tol = dict(rtol=3e-2, atol=3e-3) if autocast else dict(rtol=1e-3, atol=1e-5)
    for (name, pa), pb in zip(a.named_parameters(), b.parameters()):
        scale = float(pb.grad.abs().max())
        assert scale > 0, name
        assert torch.allclose(pa.grad / scale, pb.grad / scale, **tol), (name, float((pa.grad - pb.grad).abs().max()), scale)


def test_optimizer_step_refreshes_bf16_weights(dev):
    from refign_amd.linear import Linear
    torch.manual_seed(0)
    m = Linear(32, 16).to(dev)
    opt = torch.optim.AdamW(m.parameters(), lr=0.1, fused=True)
    x = torch.randn(64, 32, device=dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y0 = m(x)
        y0.float().sum().backward()
        opt.step()
        y1 = m(x)
        want = F.linear(x.bfloat16(), m.weight.detach().bfloat16(), m.bias.detach().bfloat16())
    assert not torch.equal(y0, y1) and torch.equal(y1, want)


def test_refresh_multi_tensor_cast_is_torch_rounding(dev):
    """refresh() of many cached bf16 copies = one multi-tensor cast launch; bit-identical to tensor.to(bfloat16)
    (round to nearest even, NaN / inf / denormals), ragged sizes, chunked large tensors, unaligned views."""
    from refign_amd.params import as_dtype, refresh
    g = torch.Generator().manual_seed(1)
    sizes = [(19,), (1,), (3, 5), (320,), (320, 320), (257, 129), (40000,), (7, 11, 13), (1280, 320), (2,)]
    flat = torch.randn(sum(int(torch.tensor(s).prod()) for s in sizes) + 3, generator=g).to(dev)
    ps, off = [], 1                                              # views at odd element offsets: 4-byte aligned only
    for s in sizes:
        n = int(torch.tensor(s).prod())
        ps.append(nn.Parameter(flat[off:off + n].view(s)))
        off += n
    ps.append(nn.Parameter(torch.tensor([float("nan"), float("inf"), -float("inf"), 1e-40, -0.0, 3.3895314e38, 1.0 + 2 ** -8,
                                         1.0 + 3 * 2 ** -9], device=dev)))
    copies = [as_dtype(p, torch.bfloat16) for p in ps]
    with torch.no_grad():
        for p in ps:
            p.data.mul_(1.7).add_(0.123)                         # through .data: version counters do not move
    refresh(ps)
    for p, c in zip(ps, copies):
        assert as_dtype(p, torch.bfloat16) is c
        want = p.detach().to(torch.bfloat16)
        assert torch.equal(c.view(torch.int16), want.view(torch.int16)) or \
            (torch.isnan(want.float()) == torch.isnan(c.float())).all() and \
            torch.equal(torch.nan_to_num(c.float(), nan=0.0), torch.nan_to_num(want.float(), nan=0.0))


def test_refresh_transposed_copies_multi_tensor(dev):
    """refresh() of the cached W^T bf16 operands (params.transposed: the input-gradient GEMMs' weights) = one multi-tensor
    transpose + cast launch over 64 x 64 tiles (ABI 3): bit-identical to p.t().to(bfloat16) for shapes that are / are not
    multiples of the tile or of the 16-byte pieces, and for a storage that is only 4-byte aligned."""
    from refign_amd.params import refresh, transposed
    g = torch.Generator().manual_seed(2)
    shapes = [(320, 320), (1280, 320), (320, 1280), (64, 64), (72, 40), (19, 150), (8, 8), (130, 67), (1, 5)]
    flat = torch.randn(sum(a * b for a, b in shapes) + 3, generator=g).to(dev)
    ps, off = [], 0
    for k, (a, b) in enumerate(shapes):
        if k == 5:
            off += 1                                             # from here on: odd element offset
        ps.append(nn.Parameter(flat[off:off + a * b].view(a, b)))
        off += a * b
    copies = [transposed(p, torch.bfloat16) for p in ps]
    for p, c in zip(ps, copies):
        assert tuple(c.shape) == (p.shape[1], p.shape[0]) and c.is_contiguous()
    with torch.no_grad():
        for p in ps:
            p.data.mul_(-0.7).add_(0.321)                        # through .data: version counters do not move
    refresh(ps)
    for p, c in zip(ps, copies):
        assert transposed(p, torch.bfloat16) is c
        assert torch.equal(c, p.detach().t().to(torch.bfloat16)), tuple(p.shape)


def test_python_side_workspace_sizes_match_the_abi(dev):
    """layernorm.py / dwconv.py size their scratch buffers without an ABI round trip: same numbers as the library."""
    from refign_amd import _lib, dwconv, layernorm
    lib = _lib.load_library()
    for C in (8, 64, 320, 1024, 1280):
        assert lib.rfn_layernorm_bwd_workspace_bytes(C) == layernorm._LN_WS_ROWS * 2 * C * 4
        assert lib.rfn_dwconv3x3_bwd_weight_workspace_bytes(C) == dwconv._DW_WS_STRIPES * 10 * C * 4


@pytest.mark.parametrize("H,W,r", [(16, 24, 2), (17, 30, 2), (34, 60, 4), (135, 50, 8), (8, 8, 8)])
@pytest.mark.parametrize("sink", [False, True])
def test_spatial_reduction_conv_as_patch_linear(dev, H, W, r, sink):
    """conv.patch_conv_tokens == the strided convolution it replaces (forward, input / weight / bias gradients),
    including maps whose size is not a multiple of the stride, with and without the flat gradient buffer."""
    from refign_amd.conv import Conv2d, patch_conv_tokens
    from refign_amd.trainer import FlatGradBuffer
    torch.manual_seed(H * 100 + r)
    C, B = 32, 2
    a, b = Conv2d(C, C, r, stride=r).to(dev), nn.Conv2d(C, C, r, stride=r).to(dev)
    b.load_state_dict(a.state_dict())
    if sink:
        FlatGradBuffer(a.parameters())
    x = torch.randn(B, H * W, C, device=dev)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    for rep in range(2):                                          # twice: accumulation into .grad
        ya = patch_conv_tokens(xa, H, W, a)
        yb = b(xb.transpose(1, 2).reshape(B, C, H, W)).flatten(2).transpose(1, 2)
        assert ya.shape == yb.shape
        assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-4)
        g = torch.randn_like(yb)
        ya.backward(g)
        yb.backward(g)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(a.weight.grad, b.weight.grad, rtol=1e-3, atol=1e-3)
    assert torch.allclose(a.bias.grad, b.bias.grad, rtol=1e-3, atol=1e-3)
    with torch.no_grad():
        assert torch.allclose(patch_conv_tokens(x, H, W, a), yb, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("H,W,r", [(16, 24, 2), (34, 60, 4), (64, 72, 8), (17, 31, 2)])
def test_spatial_reduction_conv_bf16_gradients_land_in_the_flat_buffer(dev, H, W, r):
    """under the bf16 autocast of the training step, with the flat gradient buffer: channel-major patches, the TN kernel adds
    the weight AND bias gradient straight into the (Co, C, r, r) / (Co,) views (one launch), twice (accumulation)"""
    from refign_amd.conv import Conv2d, patch_conv_tokens
    from refign_amd.trainer import FlatGradBuffer
    torch.manual_seed(H + r)
    C, B = 64, 2
    a, b = Conv2d(C, C, r, stride=r).to(dev), nn.Conv2d(C, C, r, stride=r).to(dev)
    b.load_state_dict(a.state_dict())
    FlatGradBuffer(a.parameters())
    x = torch.randn(B, H * W, C, device=dev)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    for rep in range(2):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ya = patch_conv_tokens(xa, H, W, a)
            yb = b(xb.transpose(1, 2).reshape(B, C, H, W)).flatten(2).transpose(1, 2)
        assert ya.shape == yb.shape and ya.dtype == torch.bfloat16
        assert torch.allclose(ya.float(), yb.float(), rtol=2e-2, atol=2e-2)
        g = torch.randn_like(yb)
        ya.backward(g)
        yb.backward(g)
    for got, want in ((xa.grad, xb.grad), (a.weight.grad, b.weight.grad), (a.bias.grad, b.bias.grad)):
        scale = float(want.abs().max())
        assert torch.allclose(got / scale, want / scale, rtol=3e-2, atol=1e-2), float((got - want).abs().max()) / scale


@pytest.mark.parametrize("cmajor", [False, True])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C,r", [(2, 16, 24, 64, 2), (1, 135, 50, 32, 8), (3, 17, 30, 320, 2), (2, 34, 61, 128, 4)])
def test_patchify_tokens_kernel(dev, dt, B, H, W, C, r, cmajor):
    """csrc/upcat.hip patchify_kernel / patchify_cmajor_kernel: gather to r x r patches -- (ry, rx, c) rows, or (c, ry, rx)
    rows, the layout of the convolution weight -- and scatter back == the view/permute formulation."""
    from refign_amd import conv
    x = torch.randn(B, H * W, C, device=dev).to(dt)
    Hr, Wr = H // r, W // r
    order = (0, 1, 3, 5, 2, 4) if cmajor else (0, 1, 3, 2, 4, 5)
    want = x.view(B, H, W, C)[:, :Hr * r, :Wr * r].reshape(B, Hr, r, Wr, r, C).permute(*order) \
        .reshape(B * Hr * Wr, r * r * C)
    got, hr, wr = conv._to_patches(x, H, W, r, cmajor)
    assert (hr, wr) == (Hr, Wr) and torch.equal(got, want)
    back = conv._from_patches(got, B, H, W, C, r, Hr, Wr, cmajor)
    ref = torch.zeros(B, H, W, C, device=dev, dtype=dt)
    ref[:, :Hr * r, :Wr * r] = x.view(B, H, W, C)[:, :Hr * r, :Wr * r]
    assert torch.equal(back, ref.view(B, H * W, C))


def test_multi_tensor_adamw_matches_torch_fused(dev=None):
    """refign_amd/optim.py: AdamW of a whole parameter set as one kernel launch == torch.optim.AdamW(fused=True) over 6
    steps with fresh gradients each step: 3 parameter groups (different lr / weight decay), odd sizes (unaligned tails),
    an LR schedule that changes group['lr'] every step; parameters and both moments to 2e-6 relative, `step` counters
    equal (the optimizer state a checkpoint stores stays torch's)."""
    import copy
    from refign_amd.optim import MultiTensorAdamW
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shapes = [(257, 33), (64,), (3, 3, 16, 16), (70001,), (5,), (128, 128)]
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]

    def mk(params):
        return torch.optim.AdamW([{"params": params[:2], "lr": 1e-3, "weight_decay": 0.01},
                                  {"params": params[2:4], "lr": 3e-4, "weight_decay": 0.0},
                                  {"params": params[4:], "lr": 1e-2, "weight_decay": 0.1, "betas": (0.8, 0.99)}], fused=True)
    ref, mine = mk(ps), mk(qs)
    fast = MultiTensorAdamW(mine)
    flat = torch.zeros(sum(q.numel() for q in qs), device=dev)      # gradients as persistent views, as in the trainer
    o = 0
    for q in qs:
        q.grad = flat[o:o + q.numel()].view_as(q)
        o += q.numel()
    for it in range(6):
        for p, q in zip(ps, qs):
            g = torch.randn_like(p) * (0.1 + it)
            p.grad = g.clone()
            q.grad.copy_(g)
        for opt in (ref, mine):
            for gi, grp in enumerate(opt.param_groups):
                grp["lr"] = [1e-3, 3e-4, 1e-2][gi] * (1.0 - 0.1 * it)
        ref.step()
        fast.step()
    assert fast.launches == 5                                        # torch made the first step (state creation)
    for grp in mine.param_groups:                                    # hand-over back to torch keeps counting correctly
        grp["lr"] = 1e-3
    for p, q in zip(ps, qs):
        assert float((p - q).abs().max()) <= 2e-6 * float(p.abs().max())
        for k in ("exp_avg", "exp_avg_sq"):
            a, b = ref.state[p][k], mine.state[q][k]
            assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-12
    sd = mine.state_dict()                                           # looking at the state brings the counters up to date
    assert all(float(s["step"]) == 6.0 for s in sd["state"].values())
    for p, q in zip(ps, qs):
        assert float(ref.state[p]["step"]) == float(mine.state[q]["step"]) == 6.0
    # a configuration outside plain AdamW stays on torch's step
    ams = torch.optim.AdamW([torch.nn.Parameter(torch.randn(8, device=dev))], amsgrad=True)
    ams.param_groups[0]["params"][0].grad = torch.ones(8, device=dev)
    f2 = MultiTensorAdamW(ams)
    f2.step(); f2.step()
    assert f2.launches == 0


def test_multi_tensor_adamw_follows_load_state_dict():
    """optimizer.load_state_dict() in mid-run (resume / rollback) replaces every moment tensor and the step counters: the
    one-launch AdamW must continue from the LOADED state, not from its cached pointers and its own counter."""
    import copy
    from refign_amd.optim import MultiTensorAdamW
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    shapes = [(129, 17), (33,), (4097,)]
    ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ref, mine = torch.optim.AdamW(ps, lr=1e-2, fused=True), torch.optim.AdamW(qs, lr=1e-2, fused=True)
    fast = MultiTensorAdamW(mine)
    grads = [[torch.randn_like(p) for p in ps] for _ in range(7)]

    def run(it):
        for p, q, g in zip(ps, qs, grads[it]):
            p.grad = g.clone()
            if q.grad is None:
                q.grad = g.clone()
            else:
                q.grad.copy_(g)
        ref.step()
        fast.step()
    for it in range(3):
        run(it)
    saved = (copy.deepcopy(ref.state_dict()), [p.detach().clone() for p in ps],
             copy.deepcopy(mine.state_dict()), [q.detach().clone() for q in qs])
    run(3)
    run(4)
    ref.load_state_dict(saved[0])                                       # roll both back to the state after step 3
    mine.load_state_dict(saved[2])
    with torch.no_grad():
        for p, q, a, b in zip(ps, qs, saved[1], saved[3]):
            p.copy_(a)
            q.copy_(b)
    run(5)
    run(6)
    assert fast.launches >= 5
    for p, q in zip(ps, qs):
        assert float((p - q).abs().max()) <= 2e-6 * float(p.abs().max())
        assert float(mine.state_dict()["state"][0]["step"]) == float(ref.state[ps[0]]["step"]) == 5.0
