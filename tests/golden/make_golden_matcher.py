"""tests/golden/make_golden_matcher.py -- golden vectors for matcher TRAINING, captured from the
imported reference (authoring container only): one `AlignmentModel.training_step` (models/alignment_model.py:81-146) on a
128x160 pair with a synthetic `prime` image -- the two losses, their balancing weights, the total, gradient norms of
every sub-module of the head and samples of a few gradient tensors -- plus the loss modules alone on recorded head
outputs (so that refign_amd/losses.py is checked on CPU without any kernel).

    python tests/golden/make_golden_matcher.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_import as R  # noqa: E402

R.setup()
from fill import closed_form_fill, hashed_uniform  # noqa: E402
from make_golden_modules import save, t  # noqa: E402


def batch_arrays(b, H, W):
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    trg = (hashed_uniform((b, 3, H, W), "g14/trg") * 4 - 2).astype(np.float32)
    ref = (0.8 * np.roll(trg, (2, -3), (2, 3)) + 0.2 * (hashed_uniform((b, 3, H, W), "g14/ref") * 4 - 2)).astype(np.float32)
    flow = np.empty((b, 2, H, W), np.float32)
    for s in range(b):
        # smooth synthetic flow: an affine part + a low-frequency wave, a few pixels in magnitude
        flow[s, 0] = 3.0 - 2.0 * s + 0.03 * (xx - W / 2) + 1.5 * np.sin(2 * np.pi * yy / H + 0.7 * s)
        flow[s, 1] = -2.0 + 1.0 * s - 0.02 * (yy - H / 2) + 1.2 * np.cos(2 * np.pi * xx / W - 0.3 * s)
    mask = (xx + flow[:, 0] >= 0) & (xx + flow[:, 0] <= W - 1) & (yy + flow[:, 1] >= 0) & (yy + flow[:, 1] <= H - 1)
    mask[:, :3] = False                                              # some rows that are invalid for another reason
    return trg, ref, flow, mask


def main():
    am = R.ref_module("models.alignment_model")
    lm = R.ref_module("models.losses")
    mu = R.ref_module("helpers.matching_utils")
    vggm = R.ref_module("models.backbones.vgg")
    ua = R.ref_module("models.heads.uawarpc")
    torch.manual_seed(0)
    b, H, W = 2, 128, 160
    trg, ref, flow, mask = batch_arrays(b, H, W)
    idx = [0, 1]                                                     # prime made from ref for sample 0, from trg for sample 1
    src = np.stack([(ref, trg)[k][s] for s, k in enumerate(idx)])
    with torch.no_grad():
        prime = mu.warp(t(src), t(flow)).numpy()                     # the data module's synthetic view of image i
    vgg = closed_form_fill(vggm.VGG('vgg16', out_indices=[2, 3, 4]), "alignment_backbone.").eval()
    head = closed_form_fill(ua.UAWarpCHead(in_index=[0, 1], input_transform='multiple_select',
                                           estimate_uncertainty=True)).train()
    ss = lm.MultiScaleFlowLoss(loss_type='HuberLoss')
    us = lm.WBipathLoss(objective='multi_scale_flow_loss', loss_type='HuberLoss', visibility_mask=False)
    logged, passes = {}, []

    def head_rec(*a):
        out = head(*a)
        passes.append(out)
        return out

    ns = types.SimpleNamespace(alignment_backbone=vgg, alignment_head=head_rec, selfsupervised_loss=ss,
                               unsupervised_loss=us, apply_constant_flow_weights=False,
                               weights_selfsupervised_and_unsupervised=am.AlignmentModel.weights_selfsupervised_and_unsupervised,
                               log=lambda k, v, **kw: logged.__setitem__(k, float(v)))
    batch = {"image_ref": t(ref), "image_trg": t(trg), "image_prime": t(prime), "flow_prime": t(flow),
             "mask_prime": t(mask), "prime_trg_idx": idx}
    loss = am.AlignmentModel.training_step(ns, batch, 0)
    loss.backward()
    with torch.no_grad():
        ss_loss = ss(passes[0], t(flow), mask=t(mask))
        us_loss, masks, _, comp = us(passes[1], passes[2], t(flow), mask_used=t(mask), return_masks=True)
    # (sic) the step passes apply_constant_flow_weights in the position of weight_ss: alignment_model.py:140-142
    w_ss, w_us = am.AlignmentModel.weights_selfsupervised_and_unsupervised(ss_loss, us_loss, False)
    arrays = dict(size=np.array([H, W]), image_prime=prime, flow_prime=flow, mask_prime=mask, prime_trg_idx=np.array(idx),
                  loss=np.float64(loss.item()), ss_loss=np.float64(ss_loss.item()), us_loss=np.float64(us_loss.item()),
                  weight_ss=np.float64(w_ss), weight_us=np.float64(w_us))
    for name, mod in head.named_children():
        g = [p.grad for p in mod.parameters() if p.grad is not None]
        if g:
            arrays["gradnorm/" + name] = np.float64(torch.sqrt(sum((x.double() ** 2).sum() for x in g)).item())
    for name in ("decoder4.conv_0.0.weight", "decoder1.predict_mapping.weight", "estimate_uncertainty_components1.conv_0.0.weight",
                 "refinement_module_finest.dc_conv7.weight"):
        p = dict(head.named_parameters()).get(name)
        if p is not None and p.grad is not None:
            arrays["grad/" + name] = p.grad.numpy().reshape(-1)[::7].copy()
    # the loss modules alone: recorded head outputs (all four levels of the three passes) -> losses and masks
    for pi, out in enumerate(passes):
        for li, (fl, un) in enumerate(out):
            arrays[f"pass{pi}/flow{li}"] = fl.detach().numpy()
            arrays[f"pass{pi}/uncert{li}"] = un.detach().numpy()
    for li, m in enumerate(masks):
        arrays[f"us_mask{li}"] = m.numpy()
    arrays["us_composed_flow3"] = comp[3][0].detach().numpy()
    arrays["us_composed_uncert3"] = comp[3][1].detach().numpy()
    save("matcher_step_128x160", **arrays)
    print({k: float(v) for k, v in arrays.items() if np.asarray(v).ndim == 0})
    losses_alone(lm, flow, mask, H, W)


def losses_alone(lm, flow, mask, H, W):
    """The two loss modules on synthetic four-level estimates that are CLOSE to consistent (so that the visibility mask
    of the W-bipath loss is neither empty nor full), with gradients with respect to every input."""
    b = flow.shape[0]
    sizes = [(16, 16), (32, 32), (H // 8, W // 8), (H // 4, W // 4)]
    tf = t(flow)
    first, second, req = [], [], []
    for li, (h, w) in enumerate(sizes):
        base = torch.nn.functional.interpolate(tf, (h, w), mode='bilinear', align_corners=False)
        n = lambda k, c, a: t(((hashed_uniform((b, c, h, w), f"g14/{k}{li}") - 0.5) * a).astype(np.float32))  # noqa: E731
        f1 = (0.55 * base + n("f", 2, 1.5)).requires_grad_(True)
        f2 = (0.45 * base + n("s", 2, 1.5)).requires_grad_(True)
        u1 = n("uf", 1, 2.0).requires_grad_(True)
        u2 = n("us", 1, 2.0).requires_grad_(True)
        first.append((f1, u1))
        second.append((f2, u2))
        req += [f1, u1, f2, u2]
    ss = lm.MultiScaleFlowLoss(loss_type='HuberLoss', level_weights=[0.32, 0.08, 0.02, 0.01])
    us = lm.WBipathLoss(objective='multi_scale_flow_loss', loss_type='HuberLoss', visibility_mask=True,
                        detach_flow_for_warping=True)
    us2 = lm.WBipathLoss(objective='multi_scale_flow_loss', loss_type='L1Loss', visibility_mask=False,
                         detach_flow_for_warping=False)
    arrays = {"flow_prime": flow, "mask_prime": mask}
    l_ss = ss(first, tf, mask=t(mask))
    l_us, masks, cyc, comp = us(first, second, tf, mask_used=t(mask), return_masks=True)
    (l_ss + l_us).backward()
    arrays.update(ss_loss=np.float64(l_ss.item()), us_loss=np.float64(l_us.item()))
    for li in range(4):
        for nm, x in (("f", first[li][0]), ("uf", first[li][1]), ("s", second[li][0]), ("us", second[li][1])):
            arrays[f"in/{nm}{li}"] = x.detach().numpy()
            arrays[f"grad/{nm}{li}"] = x.grad.numpy().copy()
            x.grad = None
        arrays[f"mask{li}"] = masks[li].numpy()
        arrays[f"cyclic{li}"] = cyc[li].numpy()
        arrays[f"composed{li}"] = comp[li][0].detach().numpy()
    # deterministic flows only (no uncertainty), L1, gradient THROUGH the warping flow
    l2 = us2([f for f, _ in first], [f for f, _ in second], tf, mask_used=t(mask))
    l2.backward()
    arrays["us_l1_nodetach_loss"] = np.float64(l2.item())
    for li in range(4):
        arrays[f"grad_l1/f{li}"] = first[li][0].grad.numpy().copy()
        arrays[f"grad_l1/s{li}"] = second[li][0].grad.numpy().copy()
    save("matcher_losses_128x160", **arrays)
    print("losses alone:", float(l_ss), float(l_us), float(l2), [float(m.float().mean()) for m in masks],
          [float(c.float().mean()) for c in cyc])


if __name__ == "__main__":
    main()
