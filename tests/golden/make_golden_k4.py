"""tests/golden/make_golden_k4.py -- G7-K4: the reference's align() (segmentation_model.py:493-523) at the size the metric
is quoted on, 1080 x 1920, b = 1, closed-form weights (fill.py), smooth logits (make_golden_modules.smooth_logits): the
north-star parity statement (warped logits within 1e-3, pixel-exact argmax) pinned where the benchmark runs, not only
at 128 x 160.  Run in the build container (the reference is imported from /root/reference, tests/golden/_ref_import.py):
    python tests/golden/make_golden_k4.py
Outputs are stored as strided samples + fp64 checksums (the full tensors are 158 MB): align_smooth_1080x1920.npz."""
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402
from fill import closed_form_fill, hashed_uniform  # noqa: E402
from make_golden_modules import _head, save, smooth_logits, t  # noqa: E402


def g7_k4():
    sm = R.ref_module("models.segmentation_model")
    vggm = R.ref_module("models.backbones.vgg")
    vgg = closed_form_fill(vggm.VGG('vgg16', out_indices=[2, 3, 4]), "alignment_backbone.").eval()
    head = _head()
    H, W = 1080, 1920
    img_trg = (hashed_uniform((1, 3, H, W), "g7k4/trg") * 4 - 2).astype(np.float32)
    img_ref = (0.8 * np.roll(img_trg, (2, -3), (2, 3)) + 0.2 * (hashed_uniform((1, 3, H, W), "g7k4/ref") * 4 - 2)).astype(np.float32)
    logits = smooth_logits(19, H, W, "g7k4/logits")
    ns = types.SimpleNamespace(alignment_backbone=vgg, alignment_head=head)
    t0 = time.perf_counter()
    with torch.no_grad():
        warped, mask, cert = sm.DomainAdaptationSegmentationModel.align(ns, t(logits), t(img_ref), t(img_trg))
        t1 = time.perf_counter()
        i256 = [torch.nn.functional.interpolate(t(x), size=(256, 256), mode='area') for x in (img_ref, img_trg)]
        pyr = vgg(torch.cat([t(img_ref), t(img_trg)]), extract_only_indices=[-3, -2])
        pyr256 = vgg(torch.cat(i256), extract_only_indices=[-2, -1])
        pr, pt = zip(*[torch.split(l, [1, 1]) for l in pyr])
        pr256, pt256 = zip(*[torch.split(l, [1, 1]) for l in pyr256])
        levels = head(pt, pr, pt256, pr256, (H, W))
    print(f"  reference align() at {H}x{W}: {t1 - t0:.1f} s on {torch.get_num_threads()} threads")
    wn = warped.numpy()
    srt = np.sort(wn, axis=1)
    m = mask.numpy()
    arrays = dict(size=np.array([H, W]), cpu_seconds=np.float32(t1 - t0),
                  warped_sample=wn[:, :, ::16, ::16].copy(), warped_checksum=np.float64(wn.astype(np.float64).sum()),
                  warped_abs_checksum=np.float64(np.abs(wn.astype(np.float64)).sum()),
                  warped_argmax=wn.argmax(1).astype(np.uint8)[:, ::4, ::4].copy(),
                  warped_margin=(srt[:, -1] - srt[:, -2]).astype(np.float16)[:, ::4, ::4].copy(),
                  mask_bits=np.packbits(m.reshape(-1)), mask_count=np.int64(m.sum()),
                  cert_sample=cert.numpy()[:, :, ::8, ::8].copy(), cert_checksum=np.float64(cert.numpy().astype(np.float64).sum()))
    for name, fs in (("pyr", pyr), ("pyr256", pyr256)):
        for i, f in enumerate(fs):
            f = f.numpy()
            arrays[f"{name}{i}_sample"] = f[:, ::16, ::8, ::8].copy()
            arrays[f"{name}{i}_abs_checksum"] = np.float64(np.abs(f.astype(np.float64)).sum())
    for lvl, (fl, un) in zip((4, 3, 2, 1), levels):
        fl, un = fl.numpy(), un.numpy()
        st = 4 if lvl <= 2 else 1
        arrays[f"flow{lvl}_sample"] = fl[:, :, ::st, ::st].copy()
        arrays[f"uncert{lvl}_sample"] = un[:, :, ::st, ::st].copy()
        arrays[f"flow{lvl}_abs_checksum"] = np.float64(np.abs(fl.astype(np.float64)).sum())
    save("align_smooth_1080x1920", **arrays)


def g_k2():
    """K2 (BASELINE.json config 2: "UAWarpC align-only, 512x512 pairs"): the reference's AlignmentModel.forward
    (models/alignment_model.py:55-79) on b = 2 pairs at 512 x 512, closed-form weights: flow i -> j at full resolution and
    1 - P_R.  -> alignment_forward_512x512.npz (strided samples + fp64 checksums)."""
    am = R.ref_module("models.alignment_model")
    vggm = R.ref_module("models.backbones.vgg")
    vgg = closed_form_fill(vggm.VGG('vgg16', out_indices=[2, 3, 4]), "alignment_backbone.").eval()
    head = _head()
    B, H, W = 2, 512, 512
    img_i = (hashed_uniform((B, 3, H, W), "k2/i") * 4 - 2).astype(np.float32)
    img_j = (0.8 * np.roll(img_i, (3, -2), (2, 3)) + 0.2 * (hashed_uniform((B, 3, H, W), "k2/j") * 4 - 2)).astype(np.float32)
    ns = types.SimpleNamespace(alignment_backbone=vgg, alignment_head=head)
    t0 = time.perf_counter()
    with torch.no_grad():
        flow, uncert = am.AlignmentModel.forward(ns, t(img_i), t(img_j))
    dt = time.perf_counter() - t0
    print(f"  reference AlignmentModel.forward at {B}x{H}x{W}: {dt:.1f} s on {torch.get_num_threads()} threads")
    fl, un = flow.numpy(), uncert.numpy()
    save("alignment_forward_512x512", size=np.array([B, H, W]), cpu_seconds=np.float32(dt),
         flow_sample=fl[:, :, ::4, ::4].copy(), flow_abs_checksum=np.float64(np.abs(fl.astype(np.float64)).sum()),
         uncert_sample=un[:, :, ::4, ::4].copy(), uncert_checksum=np.float64(un.astype(np.float64).sum()))


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    R.setup()
    if len(sys.argv) > 1 and sys.argv[1] == "K2":
        g_k2()
    else:
        g7_k4()
