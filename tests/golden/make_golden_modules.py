"""tests/golden/make_golden_modules.py -- golden vectors for the weight-bearing modules of the align path
(G4 decoder / refinement / uncertainty, G5 UAWarpCHead, G7 align() end to end), captured from the imported
reference with the closed-form weights of fill.py.  Called by make_golden.py."""
import os
import types

import numpy as np
import torch

import _ref_import as R
from fill import closed_form_fill, hashed_uniform

HERE = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} kB)")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def unit(shape, key):
    """unit-norm-over-channels features in float32"""
    x = hashed_uniform(shape, key) - 0.5
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def g4():
    mods = R.ref_module("models.modules")
    dec = closed_form_fill(mods.OpticalFlowEstimatorResidualConnection(84, batch_norm=True, output_x=True),
                           "decoder3.").eval()
    x = (hashed_uniform((2, 84, 12, 10), "g4/dec_in") - 0.5).astype(np.float32)
    m, f = dec(t(x))
    save("mod_decoder84", x=x, mapping=m.numpy(), feat=f.numpy())
    ref = closed_form_fill(mods.RefinementModule(32, batch_norm=True), "refinement_module_adaptive.").eval()
    x = (hashed_uniform((1, 32, 20, 24), "g4/ref_in") - 0.5).astype(np.float32)
    save("mod_refinement32", x=x, out=ref(t(x)).numpy())
    for s, prev in [(9, True), (16, False)]:
        um = closed_form_fill(mods.UncertaintyModule(1, search_size=s, feed_in_previous=prev),
                              "estimate_uncertainty_components%d." % (3 if s == 9 else 4)).eval()
        corr = np.maximum(hashed_uniform((2, s * s, 6, 7), f"g4/u{s}_corr") - 0.3, 0).astype(np.float32)
        corr /= np.maximum(np.linalg.norm(corr, axis=1, keepdims=True), 1e-12)
        feat = (hashed_uniform((2, 32, 6, 7), f"g4/u{s}_feat") - 0.5).astype(np.float32)
        if prev:
            pu = (hashed_uniform((2, 1, 6, 7), f"g4/u{s}_pu") * 4 - 2).astype(np.float32)
            pf = (hashed_uniform((2, 2, 6, 7), f"g4/u{s}_pf") * 6 - 3).astype(np.float32)
            out = um(t(corr), t(feat), t(pu), t(pf))
            save(f"mod_uncertainty{s}", corr=corr, feat=feat, prev_uncert=pu, prev_flow=pf, out=out.numpy())
        else:
            out = um(t(corr), t(feat))
            save(f"mod_uncertainty{s}", corr=corr, feat=feat, out=out.numpy())


def _head():
    ua = R.ref_module("models.heads.uawarpc")
    head = ua.UAWarpCHead(in_index=[0, 1], input_transform='multiple_select', estimate_uncertainty=True)
    return closed_form_fill(head).eval()


def g5():
    """Full head at K1 pyramid shapes (256x256 image) and at a non-square 192x320 image."""
    head = _head()
    for name, (H, W) in [("k1_256x256", (256, 256)), ("rect_192x320", (192, 320))]:
        pyr = {
            "trg": [unit((1, 128, H // 4, W // 4), f"g5/{name}/t1"), unit((1, 256, H // 8, W // 8), f"g5/{name}/t2")],
            "src": [unit((1, 128, H // 4, W // 4), f"g5/{name}/s1"), unit((1, 256, H // 8, W // 8), f"g5/{name}/s2")],
            "trg256": [unit((1, 256, 32, 32), f"g5/{name}/t3"), unit((1, 512, 16, 16), f"g5/{name}/t4")],
            "src256": [unit((1, 256, 32, 32), f"g5/{name}/s3"), unit((1, 512, 16, 16), f"g5/{name}/s4")],
        }
        # make the source a smoothed/shifted version of the target so that the correlation has structure
        for k_t, k_s in (("trg", "src"), ("trg256", "src256")):
            for i in range(2):
                a = pyr[k_t][i]
                mix = 0.7 * np.roll(a, shift=(1, -2), axis=(2, 3)) + 0.3 * pyr[k_s][i]
                pyr[k_s][i] = (mix / np.linalg.norm(mix, axis=1, keepdims=True)).astype(np.float32)
        outs = head([t(x) for x in pyr["trg"]], [t(x) for x in pyr["src"]], [t(x) for x in pyr["trg256"]],
                    [t(x) for x in pyr["src256"]], (H, W))
        arrays = {}
        for lvl, (fl, un) in zip((4, 3, 2, 1), outs):
            arrays[f"flow{lvl}"] = fl.numpy()
            arrays[f"uncert{lvl}"] = un.numpy()
        save("head_" + name, size=np.array([H, W]), **arrays)


def g7():
    """align() end to end on a 128x160 pair (VGG-16 + head + tail), and AlignmentModel.forward."""
    sm = R.ref_module("models.segmentation_model")
    am = R.ref_module("models.alignment_model")
    vggm = R.ref_module("models.backbones.vgg")
    vgg = closed_form_fill(vggm.VGG('vgg16', out_indices=[2, 3, 4]), "alignment_backbone.").eval()
    head = _head()
    H, W = 128, 160
    img_trg = (hashed_uniform((1, 3, H, W), "g7/trg") * 4 - 2).astype(np.float32)
    img_ref = (0.8 * np.roll(img_trg, (2, -3), (2, 3)) + 0.2 * (hashed_uniform((1, 3, H, W), "g7/ref") * 4 - 2)).astype(np.float32)
    logits = (hashed_uniform((1, 19, H, W), "g7/logits") * 8 - 4).astype(np.float32)
    ns = types.SimpleNamespace(alignment_backbone=vgg, alignment_head=head)
    warped, mask, cert = sm.DomainAdaptationSegmentationModel.align(ns, t(logits), t(img_ref), t(img_trg))
    flow, unc = am.AlignmentModel.forward(ns, t(img_trg), t(img_ref))
    wn = warped.numpy()
    save("align_128x160", size=np.array([H, W]), warped_sample=wn[:, :, ::2, ::2],
         warped_checksum=np.float64(wn.astype(np.float64).sum()), warped_abs_checksum=np.float64(np.abs(wn).sum()),
         warped_argmax=wn.argmax(1).astype(np.uint8), mask=mask.numpy(), cert=cert.numpy(), flow=flow.numpy(),
         uncert=unc.numpy())


def smooth_logits(C, H, W, key):
    """Low-pass logits with large top-2 margins almost everywhere: one low-frequency plane wave per class (at most one
    period over the image) plus a class offset.  |d logit / d pixel| <= 2 pi 6 / min(H, W) ~ 0.3, so a flow error of
    1e-3 px moves a bilinear sample by ~3e-4: the fixture can tell a correct warp from a slightly wrong one at the
    north-star tolerance (1e-3), which white-noise logits cannot."""
    u = hashed_uniform((C, 4), key)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64) / H, np.arange(W, dtype=np.float64) / W, indexing="ij")
    out = np.empty((1, C, H, W), np.float32)
    for c in range(C):
        fy, fx = np.round(u[c, 0] * 2 - 1, 2), np.round(u[c, 1] * 2 - 1, 2)
        out[0, c] = 6.0 * np.cos(2 * np.pi * (fy * yy + fx * xx + u[c, 2])) + 2.0 * (u[c, 3] - 0.5)
    return out


def g7s():
    """align() on the G7 image pair with SMOOTH logits, plus every intermediate of the path (feature pyramids, the four
    (flow, log-variance) levels) so that a parity failure can be bisected per stage."""
    sm = R.ref_module("models.segmentation_model")
    vggm = R.ref_module("models.backbones.vgg")
    vgg = closed_form_fill(vggm.VGG('vgg16', out_indices=[2, 3, 4]), "alignment_backbone.").eval()
    head = _head()
    H, W = 128, 160
    img_trg = (hashed_uniform((1, 3, H, W), "g7/trg") * 4 - 2).astype(np.float32)
    img_ref = (0.8 * np.roll(img_trg, (2, -3), (2, 3)) + 0.2 * (hashed_uniform((1, 3, H, W), "g7/ref") * 4 - 2)).astype(np.float32)
    logits = smooth_logits(19, H, W, "g7s/logits")
    ns = types.SimpleNamespace(alignment_backbone=vgg, alignment_head=head)
    with torch.no_grad():
        warped, mask, cert = sm.DomainAdaptationSegmentationModel.align(ns, t(logits), t(img_ref), t(img_trg))
        # intermediates, through the same public calls align() makes (segmentation_model.py:498-513)
        i256 = [torch.nn.functional.interpolate(t(x), size=(256, 256), mode='area') for x in (img_ref, img_trg)]
        pyr = vgg(torch.cat([t(img_ref), t(img_trg)]), extract_only_indices=[-3, -2])
        pyr256 = vgg(torch.cat(i256), extract_only_indices=[-2, -1])
        pr, pt = zip(*[torch.split(l, [1, 1]) for l in pyr])
        pr256, pt256 = zip(*[torch.split(l, [1, 1]) for l in pyr256])
        levels = head(pt, pr, pt256, pr256, (H, W))
    wn = warped.numpy()
    srt = np.sort(wn, axis=1)
    arrays = dict(size=np.array([H, W]), warped_sample=wn[:, :, ::2, ::2],
                  warped_checksum=np.float64(wn.astype(np.float64).sum()), warped_abs_checksum=np.float64(np.abs(wn).sum()),
                  warped_argmax=wn.argmax(1).astype(np.uint8), warped_margin=(srt[:, -1] - srt[:, -2]).astype(np.float16),
                  mask=mask.numpy(), cert=cert.numpy(), img256_trg=i256[1].numpy()[:, :, ::4, ::4])
    for name, fs in (("pyr", pyr), ("pyr256", pyr256)):
        for i, f in enumerate(fs):
            f = f.numpy()
            arrays[f"{name}{i}_sample"] = f[:, ::8, ::2, ::2]
            arrays[f"{name}{i}_abs_checksum"] = np.float64(np.abs(f).sum())
    for lvl, (fl, un) in zip((4, 3, 2, 1), levels):
        arrays[f"flow{lvl}"] = fl.numpy()
        arrays[f"uncert{lvl}"] = un.numpy()
    save("align_smooth_128x160", **arrays)


GROUPS = {"G4": g4, "G5": g5, "G7": g7, "G7s": g7s}
