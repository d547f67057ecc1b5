"""N4: the DACS data step on the HIP kernels (csrc/dacs.hip, refign_amd/dacs.py) against the torch formulation of the same
operators in refign_amd/uda.py (strong_transform / one_mix / get_class_masks: helpers/dacs_transforms.py:14-24,43-112 as
called by models/segmentation_model.py:525-582), with identical random draws.  Labels and the mixed / unmixed choice are
exact; the image agrees to float rounding (1e-5: the contrast mean is a different summation order, the blur taps are
rounded from float64)."""
import random
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(dev, B, H, W, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    src = torch.randn(B, 3, H, W, generator=g).to(dev)
    trg = torch.randn(B, 3, H, W, generator=g).to(dev)
    gt = torch.randint(0, 19, (B, H // 8, W // 8), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)
    gt[torch.rand(B, H, W, generator=g) < 0.05] = 255
    probs = torch.softmax(4 * torch.randn(B, 19, H, W, generator=g), 1).to(dev)
    return src, trg, gt.to(dev), probs


@pytest.mark.parametrize("jitter", [False, True])
@pytest.mark.parametrize("blur", [False, True])
@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 40, 72)])
def test_dacs_mix_kernels_match_torch_formulation(monkeypatch, jitter, blur, B, H, W):
    from refign_amd.uda import DomainAdaptationSegmentationModel as M
    dev = torch.device("cuda:0")
    src, trg, gt, probs = _inputs(dev, B, H, W, 3)
    ns = types.SimpleNamespace(color_jitter_s=0.2, color_jitter_p=-1.0 if jitter else 2.0, blur=True,
                               pseudo_label_threshold=0.968, psweight_ignore_top=3, psweight_ignore_bottom=5)
    coins = iter([0.5, 0.9 if blur else 0.1] * 2)
    monkeypatch.setattr(random, "uniform", lambda a, b: next(coins))
    outs = []
    for kernel in ("0", "1"):
        monkeypatch.setenv("RFN_DACS_KERNEL", kernel)
        np.random.seed(5); torch.manual_seed(7)
        outs.append(M.get_dacs_mix(ns, trg, probs, src, gt))
    (img0, lbl0, w0), (img1, lbl1, w1) = outs
    assert img1.shape == img0.shape and lbl1.dtype == lbl0.dtype and w1.shape == w0.shape
    assert torch.equal(lbl0, lbl1)
    assert torch.equal(w0.float(), w1.float())
    assert float((img0.float() - img1).abs().max()) <= 1e-5 * max(1.0, float(img0.abs().max()))
    # the mix really took pixels of both images
    if not jitter and not blur:
        from_src = (img1 == src).all(1).float().mean()
        from_trg = (img1 == trg).all(1).float().mean()
        assert 0.2 < float(from_src) < 0.8 and abs(float(from_src + from_trg) - 1.0) < 1e-6


def test_dacs_draws_follow_the_reference_order(monkeypatch):
    """Same generator states after the call with and without the kernels: the step's later draws (HRDA crops, drop-path)
    do not depend on which implementation mixed the batch."""
    from refign_amd.uda import DomainAdaptationSegmentationModel as M
    dev = torch.device("cuda:0")
    src, trg, gt, probs = _inputs(dev, 2, 64, 96, 4)
    ns = types.SimpleNamespace(color_jitter_s=0.2, color_jitter_p=0.2, blur=True, pseudo_label_threshold=0.968,
                               psweight_ignore_top=0, psweight_ignore_bottom=0)
    states = []
    for kernel in ("0", "1"):
        monkeypatch.setenv("RFN_DACS_KERNEL", kernel)
        random.seed(1); np.random.seed(2); torch.manual_seed(3)
        M.get_dacs_mix(ns, trg, probs, src, gt)
        states.append((random.random(), float(np.random.rand()), float(torch.rand(()))))
    assert states[0] == states[1]


def test_class_set_prefetch_equals_torch_unique():
    """uda._prefetch_classes / _take_class_prefetch: the class set of the next batch from a fixed-size histogram + an
    asynchronous copy (no host synchronisation in the step) == torch.unique of the labels, ignore label included."""
    from refign_amd.uda import DomainAdaptationSegmentationModel as M
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(9)
    gt = torch.tensor([0, 3, 7, 18, 255, 11])[torch.randint(0, 6, (3, 40, 56), generator=g)].to(dev)
    ns = types.SimpleNamespace()
    M._prefetch_classes(ns, gt, 2)
    got = M._take_class_prefetch(ns, gt, 2)
    assert got is not None and torch.equal(got.cpu(), torch.unique(gt[:2]).cpu())
    assert M._take_class_prefetch(ns, gt, 2) is None                      # consumed
    M._prefetch_classes(ns, gt, 2)
    assert M._take_class_prefetch(ns, gt.clone(), 2) is None              # a different tensor: not this prefetch


@pytest.mark.parametrize("jitter", [False, True])
@pytest.mark.parametrize("blur", [False, True])
def test_dacs_mix_halves_equal_the_whole(jitter, blur):
    """Round 5: the mix in two launches -- the image half before the teacher's pseudo-labels exist (the student's forward on the
    mixed image starts next to the teacher branch), labels + weights afterwards -- is bit for bit the one-launch mix."""
    from refign_amd import dacs
    dev = torch.device("cuda:0")
    B, H, W = 2, 64, 96
    src, trg, gt, probs = _inputs(dev, B, H, W, 11)
    pp, pl = torch.max(probs, 1)
    pw = torch.full_like(pp, 0.37)
    np.random.seed(5); torch.manual_seed(7)
    bits = dacs.draw_class_bits(torch.unique(gt), B)
    jit = [dacs.draw_jitter(0.2) if jitter else None for _ in range(B)]
    sig = [0.8 if blur else None for _ in range(B)]
    img, lbl, wgt = dacs.mix(src, trg, gt, pl, pw, bits, jit, sig)
    img_a, none_l, none_w = dacs.mix(src, trg, gt, None, None, bits, jit, sig, part="image")
    none_i, lbl_b, wgt_b = dacs.mix(None, None, gt, pl, pw, bits, jit, sig, part="labels")
    assert none_l is None and none_w is None and none_i is None
    assert torch.equal(img, img_a) and torch.equal(lbl, lbl_b) and torch.equal(wgt, wgt_b)
