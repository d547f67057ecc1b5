"""oracle/cpu_align.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (torch-CPU ATen ops + the correlation oracle) of the reference's align / refine orchestration, so that
(a) the UAWarpC head logic can be pinned against the G5/G7 golden vectors without a GPU and (b) bench.py's cpu_baseline
leg can time the WHOLE Refign step on the host cores.  It drives the *parameter containers* of refign_amd.align
(decoders, uncertainty modules, VGG are plain torch modules with the reference's state_dict) but none of its HIP-backed
functions: every op that is a HIP kernel in the product is a torch-CPU / oracle call here.

Follows (brdav/refign): models/heads/uawarpc.py:95-280 (head), helpers/matching_utils.py:11-57 (warp, confidence),
models/modules.py:266-274,294-333 (correlation layers), models/segmentation_model.py:438-523 (refine, align).
"""
import math

import torch
import torch.nn.functional as F


def _corr_fn_default():
    """reference correlation.cpp if oracle/_ref is built, else the C restatement"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    try:
        import build_ref
        ref = build_ref.load_prebuilt()
        if ref is not None:
            return "reference", lambda a, b: ref.forward(a.contiguous(), b.contiguous(), 1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)
    except Exception:
        pass
    import cpu_oracle
    return "port", lambda a, b: torch.from_numpy(cpu_oracle.corr_forward(a.contiguous().numpy(), b.contiguous().numpy(),
                                                                         patch_size=9))


def warp(x, flo, return_mask=False):
    """helpers/matching_utils.py:11-49 with torch-CPU grid_sample (what the reference itself calls)."""
    B, C, H, W = x.shape
    xx = torch.arange(W, dtype=flo.dtype).view(1, 1, 1, W).expand(B, 1, H, W)
    yy = torch.arange(H, dtype=flo.dtype).view(1, 1, H, 1).expand(B, 1, H, W)
    v = torch.cat((xx, yy), 1) + flo
    vx = 2.0 * v[:, 0] / max(W - 1, 1) - 1.0
    vy = 2.0 * v[:, 1] / max(H - 1, 1) - 1.0
    out = F.grid_sample(x.float(), torch.stack((vx, vy), dim=3).float(), align_corners=True, padding_mode='zeros')
    if return_mask:
        return out, (vx > -1) & (vy > -1) & (vx < 1) & (vy < 1)
    return out


def local_corr(corr_fn, feature_source, feature_target):
    """modules.py:266-274 (input1 = target, input2 = source)."""
    c = corr_fn(feature_target, feature_source)
    return F.normalize(F.relu(c.reshape(c.shape[0], 81, *c.shape[-2:])), p=2, dim=1)


def global_corr(fs, ft):
    """modules.py:294-333,361-375."""
    b, c, hs, ws = fs.shape
    _, _, ht, wt = ft.shape
    corr = torch.bmm(ft.flatten(2).transpose(1, 2), fs.flatten(2)).transpose(1, 2)          # (b, S, T)
    cb = corr / (corr.max(dim=1, keepdim=True)[0] + 1e-5)
    ca = corr / (corr.max(dim=2, keepdim=True)[0] + 1e-5)
    corr = corr * (ca * cb)
    return F.normalize(F.relu(corr.view(b, hs * ws, ht, wt)), p=2, dim=1)


def _up(x, size):
    return F.interpolate(x, size=size, mode='bilinear', align_corners=False)


@torch.no_grad()
def head_forward(head, trg, src, trg_256, src_256, out_size, corr_fn=None):
    """UAWarpCHead.forward (uawarpc.py:95-280) with estimate_uncertainty=True, on CPU tensors."""
    if corr_fn is None:
        corr_fn = _corr_fn_default()[1]
    c11, c12 = [F.normalize(t, p=2, dim=1) for t in trg]
    c13, c14 = [F.normalize(t, p=2, dim=1) for t in trg_256]
    c21, c22 = [F.normalize(t, p=2, dim=1) for t in src]
    c23, c24 = [F.normalize(t, p=2, dim=1) for t in src_256]
    H, W = out_size
    # level 4 (:111-130)
    corr4 = global_corr(c24, c14)
    est_map4, x4 = head.decoder4(corr4)
    xx = torch.arange(16, dtype=est_map4.dtype).view(1, 1, 16)
    yy = torch.arange(16, dtype=est_map4.dtype).view(1, 16, 1)
    flow4_256 = torch.stack(((est_map4[:, 0] + 1) * 15 / 2.0 - xx, (est_map4[:, 1] + 1) * 15 / 2.0 - yy), 1) * 16.0
    u4_256 = head.estimate_uncertainty_components4(corr4, x4) + 2 * math.log(16.0)

    def level(lvl, ft, fs, flow_prev, u_prev, orig, extra=None):
        h, w = ft.shape[-2:]
        scale = flow_prev.new_tensor([w / float(orig[1]), h / float(orig[0])]).view(1, 2, 1, 1)
        corr = local_corr(corr_fn, warp(fs, flow_prev * scale), ft)
        parts = [corr, flow_prev] + ([extra] if extra is not None else []) + [u_prev]
        res, x = getattr(head, f"decoder{lvl}")(torch.cat(parts, 1))
        if lvl == 3 and head.refinement_at_adaptive_res:
            res = res + head.refinement_module_adaptive(x)
        if lvl == 1 and head.refinement_at_finest_level:
            res = res + head.refinement_module_finest(x)
        return res + flow_prev, x, getattr(head, f"estimate_uncertainty_components{lvl}")(corr, x, u_prev, flow_prev)

    # level 3 (:132-173)
    flow3, x3, u3 = level(3, c13, c23, _up(flow4_256, (32, 32)), _up(u4_256, (32, 32)), (256, 256))
    flow3 = flow3 * flow3.new_tensor([W / 256.0, H / 256.0]).view(1, 2, 1, 1)
    diag = 2 * math.log(math.sqrt(H ** 2 + W ** 2) / math.sqrt(2 * 256.0 ** 2))
    u3 = u3 + diag
    # level 2 (:209-234), level 1 (:236-271)
    s2, s1 = c12.shape[-2:], c11.shape[-2:]
    flow2, x2, u2 = level(2, c12, c22, _up(flow3, s2), _up(u3, s2), (H, W))
    flow1, _, u1 = level(1, c11, c21, _up(flow2, s1), _up(u2, s1), (H, W), extra=head.reduce(_up(x2, s1)))
    flow4 = flow4_256 * flow4_256.new_tensor([W / 256.0, H / 256.0]).view(1, 2, 1, 1)
    return (flow4, u4_256 + diag), (flow3, u3), (flow2, u2), (flow1, u1)


@torch.no_grad()
def align(alignment_backbone, alignment_head, logits_ref, images_ref, images_trg, corr_fn=None):
    """segmentation_model.py:493-523."""
    b, _, h, w = images_trg.shape
    ref_256 = F.interpolate(images_ref, size=(256, 256), mode='area')
    trg_256 = F.interpolate(images_trg, size=(256, 256), mode='area')
    feats = alignment_backbone(torch.cat([images_ref, images_trg]), extract_only_indices=[-3, -2])
    feats_256 = alignment_backbone(torch.cat([ref_256, trg_256]), extract_only_indices=[-2, -1])
    pyr_ref, pyr_trg = zip(*[torch.split(f, [b, b]) for f in feats])
    pyr_ref_256, pyr_trg_256 = zip(*[torch.split(f, [b, b]) for f in feats_256])
    flow, uncert = head_forward(alignment_head, pyr_trg, pyr_ref, pyr_trg_256, pyr_ref_256, (h, w), corr_fn)[-1]
    flow, uncert = _up(flow, (h, w)), _up(uncert, (h, w))
    cert = 1.0 - torch.exp(-1.0 / (2 * torch.exp(uncert)))                      # matching_utils.py:52-57
    warped, mask = warp(logits_ref, flow, return_mask=True)
    return warped, mask, cert


@torch.no_grad()
def alignment_forward(alignment_backbone, alignment_head, images_i, images_j, corr_fn=None):
    """AlignmentModel.forward (models/alignment_model.py:55-79): flow i -> j at full resolution and 1 - P_R (K2: what
    `bench.py --workload uawarpc_align_512x512` times on the GPU and this times on the host)."""
    b, _, h, w = images_i.shape
    i_256 = F.interpolate(images_i, size=(256, 256), mode='area')
    j_256 = F.interpolate(images_j, size=(256, 256), mode='area')
    feats = alignment_backbone(torch.cat([images_j, images_i]), extract_only_indices=[-3, -2])
    feats_256 = alignment_backbone(torch.cat([j_256, i_256]), extract_only_indices=[-2, -1])
    pyr_j, pyr_i = zip(*[torch.split(f, [b, b]) for f in feats])
    pyr_j_256, pyr_i_256 = zip(*[torch.split(f, [b, b]) for f in feats_256])
    flow, uncert = head_forward(alignment_head, pyr_i, pyr_j, pyr_i_256, pyr_j_256, (h, w), corr_fn)[-1]
    flow, uncert = _up(flow, (h, w)), _up(uncert, (h, w))
    return flow, 1.0 - (1.0 - torch.exp(-1.0 / (2 * torch.exp(uncert))))       # matching_utils.py:52-57, R = 1


@torch.no_grad()
def refine(logits_trg, logits_ref, warp_mask, certs, gamma=0.25):
    """segmentation_model.py:438-482 in torch-CPU ops."""
    pt, pr = F.softmax(logits_trg, dim=1), F.softmax(logits_ref, dim=1)
    at, ar = pt.argmax(1), pr.argmax(1)
    ent = -(pt * F.log_softmax(logits_trg, dim=1)).sum(1) / math.log(logits_trg.shape[1])
    s = ent.mean(dim=(1, 2)) ** gamma
    static = torch.tensor([0, 1, 2, 3, 4, 8, 9, 10])
    M = (torch.isin(at, static) & torch.isin(ar, static)).unsqueeze(1).expand_as(pt).clone()
    M[:, 5:8] = False
    M[:, 11:] = False
    P = certs.expand_as(pt) if certs is not None else torch.full_like(pt, 0.5)
    eps = s.view(-1, 1, 1, 1) * torch.maximum(P, M.to(pt.dtype))
    if warp_mask is not None:
        eps = eps * warp_mask.unsqueeze(1)
    return (1 - eps) * pt + eps * pr
