"""GPU: the fused up-sampling + pixel-weighted cross-entropy (csrc/loss.hip, seg._UpsampleCEFn) against the unfused
formulation of the reference: F.interpolate(logits, size, bilinear, align_corners=False) -> PixelWeightedCrossEntropyLoss
(models/losses.py:10-22) -- the loss value and the gradient with respect to the LOW-resolution logits."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _reference(logits, size, target, weight, ignore):
    lg = logits.detach().clone().requires_grad_()
    up = F.interpolate(lg, size, mode="bilinear", align_corners=False)
    loss = F.cross_entropy(up.float(), target, ignore_index=ignore, reduction="none")
    if weight is not None:
        loss = loss * weight
    loss = loss.mean()
    loss.backward()
    return loss.detach(), lg.grad


@pytest.mark.parametrize("B,C,h,w,H,W", [(2, 19, 9, 13, 36, 52),        # scale 4, whole tiles and ragged ones
                                         (1, 19, 16, 20, 35, 47),       # non-integer scales >= 2, odd sizes
                                         (2, 7, 5, 8, 40, 48),          # fewer classes, scale 8 / 6
                                         (1, 19, 34, 60, 136, 240)])    # several tiles in both directions
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("weighted", [False, True])
def test_fused_upsample_ce_matches_interpolate_then_cross_entropy(dev, B, C, h, w, H, W, dtype, weighted):
    from refign_amd import seg
    g = torch.Generator().manual_seed(B * 100 + h + W)
    logits = (torch.randn(B, C, h, w, generator=g) * 3).to(dev).to(dtype)
    target = torch.randint(0, C, (B, H, W), generator=g).to(dev)
    target[torch.rand(B, H, W, generator=g).to(dev) < 0.2] = 255
    weight = torch.rand(B, H, W, generator=g).to(dev) if weighted else None
    want, want_grad = _reference(logits, (H, W), target, weight, 255)
    lg = logits.clone().requires_grad_()
    crit = seg.PixelWeightedCrossEntropyLoss(255)
    loss = crit(seg.DeferredUpsample(lg, (H, W)), target, pixel_weight=weight)
    (3.0 * loss).backward()
    assert loss.dtype == torch.float32 and lg.grad.dtype == dtype
    assert abs(float(loss.detach()) - float(want)) <= 1e-5 * max(1.0, abs(float(want)))
    # gradient: fp32 to the rounding of the sums; 16-bit: the reference's gradient is a 16-bit tensor too
    tol = 1e-5 if dtype == torch.float32 else 2.0 ** -7
    scale = float(want_grad.float().abs().max())
    assert float((lg.grad.float() / 3.0 - want_grad.float()).abs().max()) <= tol * scale


def test_all_pixels_ignored_gives_zero_loss_and_gradient(dev):
    from refign_amd import seg
    lg = torch.randn(1, 19, 8, 8, device=dev, requires_grad=True)
    target = torch.full((1, 32, 32), 255, device=dev, dtype=torch.int64)
    loss = seg.PixelWeightedCrossEntropyLoss(255)(seg.DeferredUpsample(lg, (32, 32)), target)
    loss.backward()
    assert float(loss) == 0.0 and float(lg.grad.abs().max()) == 0.0


def test_defer_logits_only_when_the_consumer_understands_it(dev):
    """The deferred (fused up-sampling + cross-entropy) form is per model / head: a module marked by a model whose
    loss is exactly PixelWeightedCrossEntropyLoss hands out DeferredUpsample, anything else gets a tensor."""
    from refign_amd import seg
    lg = torch.randn(1, 19, 8, 8, device=dev, requires_grad=True)
    assert torch.is_tensor(seg.defer_logits(lg, (32, 32)))                 # default: no fused consumer
    head, other = torch.nn.Identity(), torch.nn.Identity()

    class _Sub(seg.PixelWeightedCrossEntropyLoss):
        def forward(self, *a, **k):
            return super().forward(*a, **k)
    assert seg.mark_fused_ce_consumer(head, seg.PixelWeightedCrossEntropyLoss()) is True
    assert seg.mark_fused_ce_consumer(other, _Sub()) is False              # a subclass may override forward: tensor
    assert torch.is_tensor(seg.defer_logits(lg, (32, 32), seg.fused_ce_consumer(other)))
    assert isinstance(seg.defer_logits(lg, (32, 32), seg.fused_ce_consumer(head)), seg.DeferredUpsample)
    assert torch.is_tensor(seg.defer_logits(lg, (12, 12), True))           # scale < 2: ATen
    with torch.no_grad():
        assert torch.is_tensor(seg.defer_logits(lg, (32, 32), True))       # nothing to differentiate: ATen
