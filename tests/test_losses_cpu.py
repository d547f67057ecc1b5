"""CPU: refign_amd/losses.py (matcher-training losses) against golden vectors captured from the
reference's models/losses.py (tests/golden/make_golden_matcher.py).  MultiScaleFlowLoss needs no kernel; the W-bipath
loss warps flows and is checked on the GPU (tests/test_matcher_gpu.py)."""
import numpy as np
import torch
from conftest import golden


def _levels(z, a, b, grad=True):
    return [(torch.from_numpy(z[f"in/{a}{i}"]).requires_grad_(grad), torch.from_numpy(z[f"in/{b}{i}"]).requires_grad_(grad))
            for i in range(4)]


def test_multi_scale_flow_loss_probabilistic_huber_matches_reference():
    from refign_amd.losses import MultiScaleFlowLoss
    z = golden("matcher_losses_128x160")
    first = _levels(z, "f", "uf")
    loss = MultiScaleFlowLoss(loss_type='HuberLoss', level_weights=[0.32, 0.08, 0.02, 0.01])
    val = loss(first, torch.from_numpy(z["flow_prime"]), mask=torch.from_numpy(z["mask_prime"]))
    assert abs(float(val) - float(z["ss_loss"])) <= 1e-5 * abs(float(z["ss_loss"]))


def test_multi_scale_flow_loss_variants():
    """Deterministic L1 / L2 levels, up-sampling the estimate instead of down-sampling the ground truth, per-level masks,
    and the empty-mask case (a zero, not a NaN: losses.py:99-100)."""
    from refign_amd.losses import HuberLoss, MultiScaleFlowLoss
    torch.manual_seed(0)
    gt = torch.randn(2, 2, 32, 40) * 3
    est = [torch.randn(2, 2, 8, 10), torch.randn(2, 2, 16, 20)]
    mask = torch.rand(2, 32, 40) > 0.3
    for lt, fn in (("L1Loss", lambda a, b: (a - b).abs()), ("L2Loss", lambda a, b: (a - b) ** 2)):
        got = MultiScaleFlowLoss(loss_type=lt, level_weights=[2.0, 0.5])(est, gt, mask=mask)
        want = 0
        for w, e in zip((2.0, 0.5), est):
            g = torch.nn.functional.interpolate(gt, e.shape[-2:], mode='bilinear', align_corners=False)
            m = torch.nn.functional.interpolate(mask[:, None].float(), e.shape[-2:], mode='bilinear',
                                                align_corners=False).floor().bool()
            want = want + w * fn(e, g).sum(1, keepdim=True)[m].mean()
        assert torch.allclose(got, want, rtol=1e-6)
    up = MultiScaleFlowLoss(loss_type="L1Loss", downsample_gt_flow=False)(est[1], gt, mask=mask)
    e = torch.nn.functional.interpolate(est[1], gt.shape[-2:], mode='bilinear', align_corners=False)
    assert torch.allclose(up, (e - gt).abs().sum(1, keepdim=True)[mask[:, None]].mean(), rtol=1e-6)
    assert float(MultiScaleFlowLoss()(est, gt, mask=torch.zeros(2, 32, 40, dtype=torch.bool))) == 0.0
    x, y = torch.randn(5, 7), torch.randn(5, 7)
    assert torch.allclose(HuberLoss(delta=0.5)(x, y), 2 * 0.5 * torch.nn.functional.smooth_l1_loss(x, y, beta=0.5))


def test_loss_weighting_keeps_the_reference_quirk():
    """alignment_model.py:140-142 passes apply_constant_flow_weights where weight_ss is expected."""
    from refign_amd.alignment_model import AlignmentModel
    w = AlignmentModel.weights_selfsupervised_and_unsupervised
    z = golden("matcher_step_128x160")
    ss, us = torch.tensor(float(z["ss_loss"])), torch.tensor(float(z["us_loss"]))
    assert w(ss, us, False) == (float(z["weight_ss"]), float(z["weight_us"]))
    assert w(torch.tensor(2.0), torch.tensor(4.0)) == (2.0, 1.0) and w(torch.tensor(4.0), torch.tensor(2.0)) == (1.0, 2.0)
    assert w(torch.tensor(1.0), torch.tensor(1e-9))[1] == 100.0
    assert w(ss, us, 1.0, 1.0, True) == (1.0, 1.0)
