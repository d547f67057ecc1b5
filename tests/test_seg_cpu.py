"""Segmentation networks (MiT, DAFormer / SegFormer heads, HRDA fusion, loss) against golden vectors captured from the
imported reference.  These modules are compositions of torch ops, so the parity check runs on CPU here and again on the
GPU in test_seg_gpu.py."""
import json
import os
import random

import numpy as np
import pytest
import torch
from conftest import GOLDEN, golden
from fill import closed_form_fill, hashed_uniform

DIMS_B0 = [32, 64, 160, 256]


def img(shape, key):
    return (hashed_uniform(shape, key) * 4 - 2).astype(np.float32)


def feats(key, b, h, w, dims, dev):
    return [torch.from_numpy((hashed_uniform((b, c, h // s, w // s), f"{key}/f{i}") - 0.5).astype(np.float32)).to(dev)
            for i, (c, s) in enumerate(zip(dims, (4, 8, 16, 32)))]


def check_state_dicts():
    from refign_amd import seg
    man = json.load(open(os.path.join(GOLDEN, "state_dict_manifest.json")))
    d = [64, 128, 320, 512]
    for name, mod in [("MixVisionTransformer(mit_b5)", seg.MixVisionTransformer("mit_b5")),
                      ("MixVisionTransformer(mit_b0)", seg.MixVisionTransformer("mit_b0")),
                      ("DAFormerHead(b5)", seg.DAFormerHead(d, [0, 1, 2, 3], 19, 'multiple_select')),
                      ("SegFormerHead(b5)", seg.SegFormerHead(d, [0, 1, 2, 3], 19, 'multiple_select'))]:
        assert {k: list(v.shape) for k, v in mod.state_dict().items()} == man[name], name


@torch.no_grad()
def check_mit(dev, tol):
    from refign_amd.seg import MixVisionTransformer
    g = golden("mit_b0_64x96")
    m = closed_form_fill(MixVisionTransformer("mit_b0"), "backbone.").to(dev).eval()
    outs = m(torch.from_numpy(img((2, 3, 64, 96), "g9/b0")).to(dev))
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.cpu().numpy(), g[f"c{i + 1}"], rtol=tol, atol=tol, err_msg=f"c{i + 1}")
    g = golden("mit_b5_64x64")
    m = closed_form_fill(MixVisionTransformer("mit_b5"), "backbone.").to(dev).eval()
    outs = m(torch.from_numpy(img((1, 3, 64, 64), "g9/b5")).to(dev))
    for i, o in enumerate(outs):
        o = o.cpu().numpy()
        np.testing.assert_allclose(o[:, ::4], g[f"c{i + 1}_sample"], rtol=10 * tol, atol=10 * tol)
        assert abs(np.abs(o.astype(np.float64)).sum() - g[f"c{i + 1}_abs_sum"]) < 1e-3 * g[f"c{i + 1}_abs_sum"]


@torch.no_grad()
def check_mit_b5_k3(dev, tol, autocast=None):
    """MiT-B5 + DAFormer head at the K3/K4-shaped 136 x 240 input (odd stage maps 34x60 / 17x30 / 9x15 / 5x8, key counts
    28 / 28 / 28 / 40, batch 2) against the reference's outputs.  Returns the worst relative errors (stage maps, logits)."""
    from refign_amd.seg import DAFormerHead, MixVisionTransformer
    g = golden("mit_b5_daformer_136x240")
    dims = [64, 128, 320, 512]
    m = closed_form_fill(MixVisionTransformer("mit_b5"), "backbone.").to(dev).eval()
    head = closed_form_fill(DAFormerHead(dims, [0, 1, 2, 3], 19, 'multiple_select'), "head.").to(dev).eval()
    x = torch.from_numpy(img((2, 3, 136, 240), "g9/b5_k3")).to(dev)
    if autocast is None:
        outs = m(x)
        logits = head(outs)
    else:
        with torch.autocast("cuda", dtype=autocast):
            outs = m(x)
            logits = head(outs)
    worst = 0.0
    for i, o in enumerate(outs):
        assert tuple(o.shape) == tuple(g[f"c{i + 1}_shape"])
        o = o.float().cpu().numpy()
        want = g[f"c{i + 1}_sample"]
        err = np.abs(o[:, ::4, ::2, ::2] - want).max() / np.abs(want).max()
        worst = max(worst, err)
        assert err < tol, (f"c{i + 1}", err)
        assert abs(np.abs(o.astype(np.float64)).sum() - g[f"c{i + 1}_abs_sum"]) < max(tol, 1e-3) * g[f"c{i + 1}_abs_sum"]
    lg = logits.float().cpu().numpy()
    lerr = np.abs(lg - g["logits"]).max() / np.abs(g["logits"]).max()
    assert lerr < tol, ("logits", lerr)
    return worst, lerr, (lg.argmax(1) == g["logits"].argmax(1)).mean()


@torch.no_grad()
def check_heads(dev, tol):
    from refign_amd.seg import DAFormerHead, SegFormerHead
    f = feats("g10", 2, 64, 96, DIMS_B0, dev)
    head = closed_form_fill(DAFormerHead(DIMS_B0, [0, 1, 2, 3], 19, 'multiple_select'), "head.").to(dev).eval()
    np.testing.assert_allclose(head(f).cpu().numpy(), golden("daformer_head")["out"], rtol=tol, atol=tol)
    head = closed_form_fill(SegFormerHead(DIMS_B0, [0, 1, 2, 3], 19, 'multiple_select'),
                            "hrda_scale_attention.").to(dev).eval()
    np.testing.assert_allclose(head(f).cpu().numpy(), golden("segformer_head")["out"], rtol=tol, atol=tol)


def _build(dev):
    from refign_amd.seg import DAFormerHead, MixVisionTransformer, SegFormerHead
    bb = closed_form_fill(MixVisionTransformer("mit_b0", drop_path_rate=0.0), "backbone.").to(dev)
    hd = closed_form_fill(DAFormerHead(DIMS_B0, [0, 1, 2, 3], 19, 'multiple_select', dropout_ratio=0.0), "head.").to(dev)
    sa = closed_form_fill(SegFormerHead(DIMS_B0, [0, 1, 2, 3], 19, 'multiple_select', dropout_ratio=0.0),
                          "hrda_scale_attention.").to(dev)
    return bb, hd, sa


def check_hrda(dev, tol):
    from refign_amd.seg import hrda_backbone, hrda_head
    x = torch.from_numpy(img((2, 3, 128, 192), "g11/x")).to(dev)
    bb, hd, sa = _build(dev)
    bb.forward = hrda_backbone(bb, 4)(bb.forward)
    hd.forward = hrda_head(hd, sa, 4)(hd.forward)
    bb.train(); hd.train(); sa.train()
    random.seed(1234)
    with torch.no_grad():
        logits, hr_logits, box = hd(bb(x))
    g = golden("hrda_student")
    assert list(box) == [int(v) for v in g["crop_box"]]
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=tol, atol=tol)
    np.testing.assert_allclose(hr_logits.cpu().numpy()[:, :, ::2, ::2], g["hr_logits_sample"], rtol=tol, atol=tol)
    bb, hd, sa = _build(dev)
    bb.forward = hrda_backbone(bb, 4, is_teacher=True)(bb.forward)
    hd.forward = hrda_head(hd, sa, 4, is_teacher=True)(hd.forward)
    bb.eval(); hd.eval(); sa.eval()
    with torch.no_grad():
        out = hd(bb(x))
    np.testing.assert_allclose(out.cpu().numpy(), golden("hrda_teacher")["logits"], rtol=tol, atol=tol)


def check_loss(dev):
    from refign_amd.seg import PixelWeightedCrossEntropyLoss
    g = golden("pw_ce_loss")
    lt = torch.from_numpy(g["logits"]).to(dev).requires_grad_()
    loss = PixelWeightedCrossEntropyLoss()(lt, torch.from_numpy(g["target"]).to(dev),
                                           pixel_weight=torch.from_numpy(g["weight"]).to(dev))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    np.testing.assert_allclose(lt.grad.cpu().numpy(), g["grad"], rtol=1e-4, atol=1e-7)
    lt = torch.from_numpy(g["logits"]).to(dev).requires_grad_()
    loss = PixelWeightedCrossEntropyLoss()(lt, torch.from_numpy(g["target"]).to(dev))
    loss.backward()
    assert abs(loss.item() - float(g["loss_noweight"])) < 1e-5
    np.testing.assert_allclose(lt.grad.cpu().numpy(), g["grad_noweight"], rtol=1e-4, atol=1e-7)


CPU = torch.device("cpu")


def test_state_dicts_match_reference():
    check_state_dicts()


def test_mit_b5_daformer_k3_shape_golden_cpu():
    check_mit_b5_k3(torch.device("cpu"), 1e-3)


def test_mit_golden_cpu():
    check_mit(CPU, 2e-4)


def test_heads_golden_cpu():
    check_heads(CPU, 2e-4)


def test_hrda_golden_cpu():
    check_hrda(CPU, 5e-4)


def test_loss_golden_cpu():
    check_loss(CPU)


def test_predrawn_hrda_crop_keeps_the_python_random_stream_order():
    """seg.predraw_crop: the training step needs the adapt_to_ref coin (3rd draw of the `random` stream) before the
    source forward makes the two crop draws; pre-drawing them must give the same offsets AND the same coin."""
    import random
    import torch
    from refign_amd import seg
    x = torch.zeros(1, 3, 96, 160)
    size, div = (48, 80), 8.0
    random.seed(123)
    crop_a, box_a = seg.extract_crop(x, size, div)
    coin_a, next_a = random.random(), random.random()
    random.seed(123)
    seg.predraw_crop(96, 160, size, div)
    coin_b = random.random()                      # drawn BEFORE the crop is taken
    crop_b, box_b = seg.extract_crop(x, size, div)
    next_b = random.random()
    assert box_a == box_b and coin_a == coin_b and next_a == next_b
    assert not seg._PREDRAWN_CROPS
    # crop == image: no draw happens in either order
    random.seed(5)
    seg.predraw_crop(96, 160, (96, 160), div)
    c1 = random.random()
    assert seg.extract_crop(x, (96, 160), div) == (0, 96, 0, 160)
    random.seed(5)
    assert random.random() == c1


def test_rejoin_is_a_view_of_the_split_batch_and_a_copy_otherwise():
    """seg._rejoin (hrda_head): halves of one tensor split along dim 0 come back as a view of it -- contiguous and
    channels-last -- when no gradient is wanted; torch.cat in every other case."""
    import torch
    from refign_amd.seg import _rejoin
    for fmt in (torch.contiguous_format, torch.channels_last):
        x = torch.randn(5, 6, 4, 3).contiguous(memory_format=fmt)
        a, b = torch.split(x, [2, 3])
        j = _rejoin(a, b)
        assert torch.equal(j, x) and j.data_ptr() == x.data_ptr() and j.stride() == x.stride()
    x = torch.randn(5, 6, 4, 3)
    a, b = torch.split(x, [2, 3])
    assert _rejoin(b, a).data_ptr() != x.data_ptr() and torch.equal(_rejoin(b, a), torch.cat((b, a)))      # wrong order
    assert _rejoin(a, b.clone()).data_ptr() != x.data_ptr()                                               # other buffer
    assert _rejoin(x[:2, :3], x[2:, :3]).data_ptr() != x.data_ptr()                                       # gaps inside
    xg = x.clone().requires_grad_()
    ag, bg = torch.split(xg * 1.0, [2, 3])
    j = _rejoin(ag, bg)
    assert j.grad_fn is not None and "Cat" in type(j.grad_fn).__name__                                    # differentiated: cat


def test_deferred_upsample_materialises_for_any_other_consumer_and_flag_is_per_model():
    """the 'consumer is the fused loss' switch is an attribute of the owning model / head (not a module global: the
    last model constructed used to win for every head of the process), and a DeferredUpsample that reaches anything but the
    fused loss -- a user's own F.cross_entropy, a subclass overriding forward -- behaves as the tensor the reference passes."""
    import torch.nn as nn
    import torch.nn.functional as F
    from refign_amd import seg
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(2, 19, 6, 8, generator=g)
    target = torch.randint(0, 19, (2, 24, 32), generator=g)
    d = seg.DeferredUpsample(logits, (24, 32))
    want = F.cross_entropy(F.interpolate(logits, (24, 32), mode="bilinear", align_corners=False), target)
    assert torch.allclose(F.cross_entropy(d, target), want)
    assert tuple(d.shape) == (2, 19, 24, 32)

    class Sub(seg.PixelWeightedCrossEntropyLoss):
        def forward(self, input, target, pixel_weight=None):
            return F.cross_entropy(input, target)

    a, b = nn.Module(), nn.Module()
    assert seg.mark_fused_ce_consumer(a, seg.PixelWeightedCrossEntropyLoss()) is True
    assert seg.mark_fused_ce_consumer(b, Sub()) is False          # a subclass may not understand a DeferredUpsample
    assert seg.fused_ce_consumer(a) and not seg.fused_ce_consumer(b) and not seg.fused_ce_consumer(nn.Module())
    assert not hasattr(seg, "FUSED_CE_CONSUMER")


def test_dacs_class_bits_are_a_set_not_a_sum():
    """class ids >= 31 clamp onto bit 31; duplicates must OR (a sum carries into bit 32 and truncates), and label
    sets of more than 31 classes do not take the kernel path at all."""
    import numpy as np
    from refign_amd import dacs
    np.random.seed(0)
    classes = torch.tensor([0, 3, 31, 40, 255, 77])
    bits = dacs.draw_class_bits(classes, 4)
    assert int(bits.max()) < (1 << 32) and int(bits.min()) >= 0
    for v in bits.tolist():
        assert v & ~((1 << 0) | (1 << 3) | (1 << 31)) == 0
    img = torch.zeros(2, 3, 32, 32)
    assert not dacs.usable(img, img, torch.zeros(2, 32, 32, dtype=torch.long), num_classes=40)
