"""tests/golden/make_golden_seg.py -- golden vectors for the segmentation networks (G9 MiT, G10 decode heads,
G11 HRDA student/teacher fusion, G12 loss), captured from the imported reference with closed-form weights."""
import os
import random

import numpy as np
import torch

import _ref_import as R
from fill import closed_form_fill, hashed_uniform

HERE = os.path.dirname(os.path.abspath(__file__))
IN_CH = {"mit_b0": [32, 64, 160, 256], "mit_b5": [64, 128, 320, 512]}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} kB)")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def img(shape, key):
    return (hashed_uniform(shape, key) * 4 - 2).astype(np.float32)


def g9():
    mt = R.ref_module("models.backbones.mix_transformer")
    m = closed_form_fill(mt.MixVisionTransformer("mit_b0"), "backbone.").eval()
    x = img((2, 3, 64, 96), "g9/b0")
    outs = m(t(x))
    save("mit_b0_64x96", **{f"c{i + 1}": o.numpy() for i, o in enumerate(outs)})
    m = closed_form_fill(mt.MixVisionTransformer("mit_b5"), "backbone.").eval()
    x = img((1, 3, 64, 64), "g9/b5")
    outs = m(t(x))
    save("mit_b5_64x64", **{f"c{i + 1}_sample": o.numpy()[:, ::4] for i, o in enumerate(outs)},
         **{f"c{i + 1}_abs_sum": np.float64(o.numpy().astype(np.float64).__abs__().sum()) for i, o in enumerate(outs)})


def _feats(key, b, h, w, dims):
    return [(hashed_uniform((b, c, h // s, w // s), f"{key}/f{i}") - 0.5).astype(np.float32)
            for i, (c, s) in enumerate(zip(dims, (4, 8, 16, 32)))]


def g10():
    df = R.ref_module("models.heads.daformer")
    sf = R.ref_module("models.heads.segformer")
    dims = IN_CH["mit_b0"]
    feats = _feats("g10", 2, 64, 96, dims)
    head = closed_form_fill(df.DAFormerHead(dims, [0, 1, 2, 3], 19, 'multiple_select'), "head.").eval()
    save("daformer_head", out=head([t(f) for f in feats]).numpy())
    head = closed_form_fill(sf.SegFormerHead(dims, [0, 1, 2, 3], 19, 'multiple_select'), "hrda_scale_attention.").eval()
    save("segformer_head", out=head([t(f) for f in feats]).numpy())


def g11():
    """HRDA student (train mode, drop rates 0, seeded crop) and teacher (sliding crops) with mit_b0."""
    mt = R.ref_module("models.backbones.mix_transformer")
    df = R.ref_module("models.heads.daformer")
    sf = R.ref_module("models.heads.segformer")
    hr = R.ref_module("models.hrda")
    dims = IN_CH["mit_b0"]

    def build():
        bb = closed_form_fill(mt.MixVisionTransformer("mit_b0", drop_path_rate=0.0), "backbone.")
        hd = closed_form_fill(df.DAFormerHead(dims, [0, 1, 2, 3], 19, 'multiple_select', dropout_ratio=0.0), "head.")
        sa = closed_form_fill(sf.SegFormerHead(dims, [0, 1, 2, 3], 19, 'multiple_select', dropout_ratio=0.0),
                              "hrda_scale_attention.")
        return bb, hd, sa

    x = img((2, 3, 128, 192), "g11/x")
    bb, hd, sa = build()
    bb.forward = hr.hrda_backbone(bb, 4)(bb.forward)
    hd.forward = hr.hrda_head(hd, sa, 4)(hd.forward)
    bb.train(); hd.train(); sa.train()
    random.seed(1234)
    logits, hr_logits, box = hd(bb(t(x)))
    save("hrda_student", logits=logits.detach().numpy(), hr_logits_sample=hr_logits.detach().numpy()[:, :, ::2, ::2], crop_box=np.array(box))
    bb, hd, sa = build()
    bb.forward = hr.hrda_backbone(bb, 4, is_teacher=True)(bb.forward)
    hd.forward = hr.hrda_head(hd, sa, 4, is_teacher=True)(hd.forward)
    bb.eval(); hd.eval(); sa.eval()
    save("hrda_teacher", logits=hd(bb(t(x))).detach().numpy())


def g12():
    ls = R.ref_module("models.losses")
    rng = np.random.default_rng(12)
    logits = rng.standard_normal((2, 19, 12, 14)).astype(np.float32)
    tgt = rng.integers(0, 19, (2, 12, 14)).astype(np.int64)
    tgt[rng.random((2, 12, 14)) < 0.1] = 255
    w = rng.random((2, 12, 14)).astype(np.float32)
    lt = t(logits).requires_grad_()
    with torch.enable_grad():
        loss = ls.PixelWeightedCrossEntropyLoss()(lt, t(tgt), pixel_weight=t(w))
        loss.backward()
        lt2 = t(logits).requires_grad_()
        loss2 = ls.PixelWeightedCrossEntropyLoss()(lt2, t(tgt))
        loss2.backward()
    save("pw_ce_loss", logits=logits, target=tgt, weight=w, loss=loss.item(), grad=lt.grad.numpy(),
         loss_noweight=loss2.item(), grad_noweight=lt2.grad.numpy())


def g9k3():
    """MiT-B5 + DAFormer head on a K3/K4-SHAPED input: 136 x 240 keeps the odd geometry of the 540 x 960 views (stage maps
    34x60 / 17x30 / 9x15 / 5x8: odd sizes, ceil-division strides, key counts 4x7=28 / 4x7 / 4x7 / 40 that are not
    multiples of the 32-key attention block, ragged query tiles), batch 2.  Stored: every stage map strided + its abs
    sum, and the full 19-class logits of the head."""
    mt = R.ref_module("models.backbones.mix_transformer")
    df = R.ref_module("models.heads.daformer")
    dims = IN_CH["mit_b5"]
    m = closed_form_fill(mt.MixVisionTransformer("mit_b5"), "backbone.").eval()
    head = closed_form_fill(df.DAFormerHead(dims, [0, 1, 2, 3], 19, 'multiple_select'), "head.").eval()
    x = img((2, 3, 136, 240), "g9/b5_k3")
    outs = m(t(x))
    logits = head(outs)
    save("mit_b5_daformer_136x240",
         **{f"c{i + 1}_sample": o.numpy()[:, ::4, ::2, ::2] for i, o in enumerate(outs)},
         **{f"c{i + 1}_abs_sum": np.float64(np.abs(o.numpy().astype(np.float64)).sum()) for i, o in enumerate(outs)},
         **{f"c{i + 1}_shape": np.array(o.shape) for i, o in enumerate(outs)}, logits=logits.numpy())


GROUPS = {"G9": g9, "G9k3": g9k3, "G10": g10, "G11": g11, "G12": g12}
