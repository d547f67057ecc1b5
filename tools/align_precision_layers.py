#!/usr/bin/env python3
"""tools/align_precision_layers.py -- WHERE the timed precision map of align() loses its accuracy at 1080 x 1920.

The G7-K4 golden (tests/golden/align_smooth_1080x1920.npz: the reference's fp32 CPU output) holds the pyramids, the four
flows and the warped logits.  align() is run under the bench's autocast region with chosen sub-modules put back to fp32
(autocast off + fp32 inputs: they then take the split-bf16 fp32 path), and every variant reports, against the golden: VGG
pyramid error, flow error per level (px), warped-logit max / mean error inside the mask, confidence error, and its time.
    python tools/align_precision_layers.py [variant ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from fill import closed_form_fill, hashed_uniform  # noqa: E402
from refign_amd import align as A  # noqa: E402
from refign_amd import matching  # noqa: E402


def smooth_logits(C, H, W, key):
    """tests/golden/make_golden_modules.py::smooth_logits (low-frequency plane wave per class)"""
    u = hashed_uniform((C, 4), key)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64) / H, np.arange(W, dtype=np.float64) / W, indexing="ij")
    out = np.empty((1, C, H, W), np.float32)
    for c in range(C):
        fy, fx = np.round(u[c, 0] * 2 - 1, 2), np.round(u[c, 1] * 2 - 1, 2)
        out[0, c] = 6.0 * np.cos(2 * np.pi * (fy * yy + fx * xx + u[c, 2])) + 2.0 * (u[c, 3] - 0.5)
    return out


def fp32ify(mod):
    orig = mod.forward

    def f32(t):
        if torch.is_tensor(t):
            return t.float()
        if isinstance(t, (list, tuple)):
            return type(t)(f32(u) for u in t)
        return t

    def fwd(*a, **k):
        with torch.autocast("cuda", enabled=False):
            return orig(*[f32(t) for t in a], **{n: f32(t) for n, t in k.items()})
    mod.forward = fwd
    return lambda: setattr(mod, "forward", orig)


def main():
    dev = torch.device("cuda:0")
    with np.load(os.path.join(ROOT, "tests", "golden", "align_smooth_1080x1920.npz")) as z:
        g = {k: z[k] for k in z.files}
    H, W = [int(v) for v in g["size"]]
    vgg = closed_form_fill(A.VGG('vgg16', out_indices=[2, 3, 4]), "alignment_backbone.").to(dev).eval()
    head = closed_form_fill(A.UAWarpCHead(in_index=[0, 1], input_transform='multiple_select',
                                          estimate_uncertainty=True)).to(dev).eval()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    img_trg = (hashed_uniform((1, 3, H, W), "g7k4/trg") * 4 - 2).astype(np.float32)
    img_ref = (0.8 * np.roll(img_trg, (2, -3), (2, 3)) + 0.2 * (hashed_uniform((1, 3, H, W), "g7k4/ref") * 4 - 2)).astype(np.float32)
    logits = T(smooth_logits(19, H, W, "g7k4/logits"))
    it, ir = T(img_trg), T(img_ref)
    want_mask = np.unpackbits(g["mask_bits"])[: H * W].reshape(1, H, W).astype(bool)

    feats = vgg.features
    convs = [i for i, m in enumerate(feats) if isinstance(m, torch.nn.Conv2d)]
    variants = {
        "timed map (round 6: VGG fp16, head split-bf16)": "split",
        "fp16 everything (round 5's timed map)": lambda: [],
        "fp32 everything": None,
        "VGG fp32": lambda: [vgg],
        "head fp32": lambda: [head],
        "VGG convs 1-2 fp32": lambda: [feats[i] for i in convs[:2]],
        "VGG convs 1-4 fp32": lambda: [feats[i] for i in convs[:4]],
        "level 1 (decoder1 + finest refinement + reduce) fp32": lambda: [head.decoder1, head.refinement_module_finest, head.reduce],
        "levels 1-2 fp32": lambda: [head.decoder1, head.refinement_module_finest, head.reduce, head.decoder2],
        "flow regressors fp32 (predict_mapping x4, last refinement convs)": lambda: [
            head.decoder4.predict_mapping, head.decoder3.predict_mapping, head.decoder2.predict_mapping,
            head.decoder1.predict_mapping, head.refinement_module_adaptive.dc_convs[-1], head.refinement_module_finest.dc_convs[-1]],
        "level-1 flow regressors fp32": lambda: [head.decoder1.predict_mapping, head.refinement_module_finest.dc_convs[-1]],
        "first convolutions of decoders 3 / 2 / 1 fp32 (the ones that see the flow)": lambda: [head.decoder3.conv_0, head.decoder2.conv_0,
                                                                                              head.decoder1.conv_0],
        "first convolutions + uncertainty pred_conv_0 x3 fp32": lambda: [
            head.decoder3.conv_0, head.decoder2.conv_0, head.decoder1.conv_0] + [
            getattr(head, f"estimate_uncertainty_components{l}").pred_conv_0 for l in (3, 2, 1)],
        "first convolutions + flow regressors fp32": lambda: [
            head.decoder3.conv_0, head.decoder2.conv_0, head.decoder1.conv_0,
            head.decoder4.predict_mapping, head.decoder3.predict_mapping, head.decoder2.predict_mapping,
            head.decoder1.predict_mapping, head.refinement_module_adaptive.dc_convs[-1], head.refinement_module_finest.dc_convs[-1]],
        "decoders fp32, refinement + uncertainty fp16": lambda: [head.decoder4, head.decoder3, head.decoder2, head.decoder1, head.reduce],
        "decoders + refinement fp32, uncertainty fp16": lambda: [head.decoder4, head.decoder3, head.decoder2, head.decoder1, head.reduce,
                                                                 head.refinement_module_adaptive, head.refinement_module_finest],
        "decoder1 fp32": lambda: [head.decoder1],
        "finest refinement fp32": lambda: [head.refinement_module_finest],
        "uncertainty back ends fp32": lambda: [getattr(head, f"estimate_uncertainty_components{l}") for l in (4, 3, 2, 1)],
        "levels 3-4 (256-space) fp32": lambda: [head.decoder4, head.decoder3, head.refinement_module_adaptive],
    }
    pick = sys.argv[1:]
    for name, mods in variants.items():
        if pick and not any(p in name for p in pick):
            continue
        undo = []
        full32 = mods is None
        A.HEAD_SPLIT = mods == "split"
        if not full32 and mods != "split":
            undo = [fp32ify(m) for m in mods()]

        def run():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=not full32):
                dt = A.align_compute_dtype()
                with torch.autocast("cuda", enabled=dt != torch.float32, dtype=dt if dt != torch.float32 else None):
                    pyr = A.extract_pyramids(vgg, ir.float(), it.float())
                    levels = A.run_head(head, pyr, (H, W))
                    flow_q, logvar_q = levels[-1]
                    out = matching.align_tail(logits, flow_q.float(), logvar_q.float())
            return pyr, levels, out
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            pyr, levels, (warped, mask, cert) = run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        for u in undo:
            u()
        m = mask.cpu().numpy().astype(bool)
        both = (want_mask & m)[:, None, ::16, ::16]
        err = np.abs(warped[:, :, ::16, ::16].float().cpu().numpy() - g["warped_sample"]) * both
        cert_err = float(np.abs(cert.float().cpu().numpy()[:, :, ::8, ::8] - g["cert_sample"]).max())
        # pyramid of the full-resolution images: (ref, trg) per level -> the golden holds cat(ref, trg)
        pt, pr, pt256, pr256 = pyr
        pe = []
        for i in range(2):
            f = torch.cat([pr[i], pt[i]]).float().cpu().numpy()[:, ::16, ::8, ::8]
            pe.append(float(np.abs(f - g[f"pyr{i}_sample"]).max() / np.abs(g[f"pyr{i}_sample"]).max()))
        fe = []
        for lvl, (fl, un) in zip((4, 3, 2, 1), levels):
            st = 4 if lvl <= 2 else 1
            fe.append(float(np.abs(fl.float().cpu().numpy()[:, :, ::st, ::st] - g[f"flow{lvl}_sample"]).max()))
        ue = float(np.abs(levels[-1][1].float().cpu().numpy()[:, :, ::4, ::4] - g["uncert1_sample"]).max())
        print(f"{name:68s} {ms:7.2f} ms | pyramid rel {pe[0]:.1e} {pe[1]:.1e} | flow px L4 {fe[0]:.2e} L3 {fe[1]:.2e} L2 {fe[2]:.2e} "
              f"L1 {fe[3]:.2e} | logvar1 {ue:.2e} | warped max {err.max():.2e} mean {err.sum() / max(both.sum() * 19, 1):.2e} | "
              f"confidence {cert_err:.2e} | mask mismatch {float((m != want_mask).mean()):.1e}", flush=True)


if __name__ == "__main__":
    main()
