"""GPU: hand-written LayerNorm (csrc/layernorm.hip) forward/backward against torch's fp32 layer_norm, for the MiT
channel widths, fp32 and bf16 activations, ragged row counts."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,C", [(1, 64), (7, 32), (1000, 128), (4097, 320), (513, 512), (33, 1024), (5, 160)])
@pytest.mark.parametrize("in_dt,out_dt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                          (torch.bfloat16, torch.bfloat16)])
def test_layernorm_fwd_bwd(dev, rows, C, in_dt, out_dt):
    from refign_amd.layernorm import layer_norm
    g = torch.Generator().manual_seed(rows + C)
    x = (2 * torch.randn(rows, C, generator=g) + 0.5).to(dev).to(in_dt).requires_grad_()
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(dev).requires_grad_()
    b = (0.1 * torch.randn(C, generator=g)).to(dev).requires_grad_()
    gy = torch.randn(rows, C, generator=g).to(dev).to(out_dt)
    y = layer_norm(x, w, b, 1e-6, out_dt)
    assert y.dtype == out_dt
    y.backward(gy)
    xr = x.detach().float().requires_grad_()
    wr, br = w.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    yr = F.layer_norm(xr, (C,), wr, br, 1e-6)
    yr.backward(gy.float())
    lo = out_dt == torch.bfloat16 or in_dt == torch.bfloat16
    tol = dict(rtol=2e-2, atol=2e-2) if lo else dict(rtol=1e-4, atol=1e-5)
    assert torch.allclose(y.float(), yr, **tol)
    assert x.grad.dtype == in_dt
    assert torch.allclose(x.grad.float(), xr.grad, **(dict(rtol=2e-2, atol=2e-2) if lo else dict(rtol=1e-3, atol=1e-4)))
    rt = rows ** 0.5
    assert torch.allclose(w.grad, wr.grad, rtol=2e-2 if lo else 1e-3, atol=(2e-2 if lo else 1e-4) * rt)
    assert torch.allclose(b.grad, br.grad, rtol=2e-2 if lo else 1e-3, atol=(2e-2 if lo else 1e-4) * rt)


def test_layernorm_module_autocast_outputs_bf16(dev):
    from refign_amd.layernorm import LayerNorm
    m = LayerNorm(64, eps=1e-6).to(dev)
    x = torch.randn(3, 10, 64, device=dev)
    assert m(x).dtype == torch.float32
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert m(x).dtype == torch.bfloat16


def test_linear_split_weight_grad_matches_nn_linear(dev):
    """refign_amd.linear.Linear in fp32 (split-bf16 products on the matrix-core kernels, refign_amd/split32.py): forward
    and gradients equal to nn.Linear's to 2^-15 of the result's scale, for a token count that triggers the split-T
    weight-gradient kernel"""
    import torch.nn as nn
    from refign_amd.linear import Linear
    torch.manual_seed(0)
    a, b = Linear(96, 160).to(dev), nn.Linear(96, 160).to(dev)
    b.load_state_dict(a.state_dict())
    x = torch.randn(4, 2040, 96, device=dev)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya, yb = a(xa), b(xb)
    assert float((ya - yb).abs().max()) <= 2.0 ** -15 * float(yb.abs().max())
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(a.weight.grad, b.weight.grad, rtol=1e-3, atol=1e-2)
    assert torch.allclose(a.bias.grad, b.bias.grad, rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [64, 320, 512])
def test_layernorm_pass_through_sums_both_gradients(dev, dtype, C):
    """layernorm.layer_norm_pass: (LayerNorm(x), x) whose backward sums the gradient through the norm and the gradient of the
    pass-through output INSIDE the LayerNorm-backward kernel (rfn_layernorm_bwd_add) == autograd's own sum."""
    from refign_amd.layernorm import LayerNorm, layer_norm_pass
    torch.manual_seed(C)
    ln = LayerNorm(C, eps=1e-6).to(dev)
    x = torch.randn(3, 257, C, device=dev).to(dtype)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    g1, g2 = torch.randn_like(x), torch.randn_like(x)
    y, xp = layer_norm_pass(xa, ln.weight, ln.bias, ln.eps)
    assert xp.data_ptr() == xa.data_ptr()
    torch.autograd.backward([y, xp], [g1, g2])
    wg, bg = ln.weight.grad.clone(), ln.bias.grad.clone()
    ln.zero_grad()
    yb = ln(xb)
    torch.autograd.backward([yb, xb * 1.0], [g1, g2])
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    assert torch.equal(y, yb)
    assert float((xa.grad.float() - xb.grad.float()).abs().max()) <= tol * float(xb.grad.float().abs().max())
    assert torch.allclose(wg, ln.weight.grad, rtol=1e-4, atol=1e-4) and torch.allclose(bg, ln.bias.grad, rtol=1e-4, atol=1e-4)
    # only the pass-through output used: the gradient passes through untouched
    xc = x.clone().requires_grad_()
    _, xp = layer_norm_pass(xc, ln.weight, ln.bias, ln.eps)
    xp.backward(g2)
    assert torch.equal(xc.grad, g2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [64, 320, 512])
def test_layernorm_two_consumers_and_pass_through_sum_in_the_kernel(dev, dtype, C):
    """layernorm.layer_norm_pass2: two handles on LayerNorm(x) (q projection and key / value path of a MiT attention block) and
    the pass-through of x; the three gradients meet in ONE kernel (rfn_layernorm_bwd_add2) == autograd's own sums, also when a
    consumer or the pass-through is not differentiated."""
    from refign_amd.layernorm import LayerNorm, layer_norm_pass2
    torch.manual_seed(C + 1)
    ln = LayerNorm(C, eps=1e-6).to(dev)
    x = torch.randn(2, 301, C, device=dev).to(dtype)
    g1, g2, g3 = torch.randn_like(x), torch.randn_like(x), torch.randn_like(x)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    for use in ((1, 1, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1)):
        xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
        ln.zero_grad()
        y, y2, xp = layer_norm_pass2(xa, ln.weight, ln.bias, ln.eps)
        assert y2.data_ptr() == y.data_ptr() and xp.data_ptr() == xa.data_ptr()
        outs, grads = zip(*[(o, g) for o, g, u in zip((y, y2, xp), (g1, g2, g3), use) if u])
        torch.autograd.backward(outs, grads)
        wg, bg = ln.weight.grad.clone(), ln.bias.grad.clone()
        ln.zero_grad()
        yb = ln(xb)
        gsum = (g1.float() * use[0] + g2.float() * use[1]).to(dtype)
        torch.autograd.backward([yb] + ([xb * 1.0] if use[2] else []), [gsum] + ([g3] if use[2] else []))
        assert torch.equal(y, yb)
        assert float((xa.grad.float() - xb.grad.float()).abs().max()) <= tol * float(xb.grad.float().abs().max()), use
        assert torch.allclose(wg, ln.weight.grad, rtol=2e-2 if dtype == torch.bfloat16 else 1e-4, atol=0.3 if dtype == torch.bfloat16 else 1e-3), use
        assert torch.allclose(bg, ln.bias.grad, rtol=2e-2 if dtype == torch.bfloat16 else 1e-4, atol=0.3 if dtype == torch.bfloat16 else 1e-3), use
