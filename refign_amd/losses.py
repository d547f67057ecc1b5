"""Host mirror of the matcher-training losses of models/losses.py: same class names,
constructor keywords and forward signatures as the reference, so `configs/megadepth/*.yaml` build unmodified.

  HuberLoss            losses.py:25-34    2 delta smooth_l1(beta = delta)  ("factor 2 so it makes sense in [the]
                                          probabilistic setup")
  MultiScaleFlowLoss   losses.py:37-188   per pyramid level: (robust) end-point loss against the ground-truth flow, as a
                                          Gaussian / two-component negative log-likelihood when the level carries
                                          log-variances; masked mean; weighted sum over the levels
  WBipathLoss          losses.py:191-328  W-bipath objective: flow(prime -> source) composed with the warped
                                          flow(source -> target) must equal the synthetic flow prime -> target

Plain tensor programs on top of `matching.warp` (hand-written forward and backward kernels, csrc/warp.hip); everything
here is differentiable.  Level flows are in full-resolution pixel units at every level (heads/uawarpc.py:95-280)."""
import math
from collections.abc import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from .matching import get_gt_correspondence_mask, warp


class HuberLoss(nn.Module):
    def __init__(self, reduction='mean', delta=1.0):
        super().__init__()
        self.reduction, self.delta = reduction, delta

    def forward(self, input, target):
        return 2.0 * F.smooth_l1_loss(input, target, reduction=self.reduction, beta=self.delta) * self.delta


def _resize(x, size):
    return F.interpolate(x, size, mode='bilinear', align_corners=False)


def _level_mask(mask, size):
    """(b,H,W) bool -> (b,1,h,w) bool; a pixel of a coarser level is valid when ALL the pixels it interpolates are
    (bilinear resize of the 0/1 mask, floor: losses.py:95-99)."""
    mask = mask.unsqueeze(1)
    if tuple(mask.shape[-2:]) != tuple(size):
        mask = _resize(mask.float(), size).floor().bool()
    return mask


class MultiScaleFlowLoss(nn.Module):
    def __init__(self, level_weights=None, loss_type='L1Loss', downsample_gt_flow=True, reduction='mean'):
        super().__init__()
        self.level_weights, self.downsample_gt_flow = level_weights, downsample_gt_flow
        self.reduction, self.loss_type = reduction, loss_type
        if loss_type == 'L1Loss':
            self.loss_function = nn.L1Loss(reduction='none')
        elif loss_type == 'L2Loss':
            self.loss_function = nn.MSELoss(reduction='none')
        elif loss_type == 'HuberLoss':
            self.loss_function = HuberLoss(reduction='none')
        else:
            raise ValueError(loss_type)
        if reduction != 'mean':
            raise ValueError(reduction)

    def _one_level(self, est_flow, est_uncert, gt_flow, mask):
        if self.downsample_gt_flow:
            size = est_flow.shape[-2:]
            gt_flow = _resize(gt_flow, size)                 # values stay full-resolution pixels, like the estimates
        else:
            size = gt_flow.shape[-2:]
            est_flow = _resize(est_flow, size)
            est_uncert = None if est_uncert is None else _resize(est_uncert, size)
        if mask is not None:
            mask = _level_mask(mask, size)
            if not torch.any(mask):                          # (host synchronisation, as in the reference)
                return est_flow.new_zeros([])
        loss = self.loss_function(est_flow, gt_flow).sum(1, keepdim=True)
        if est_uncert is not None:
            # negative log-likelihood of a 2-D isotropic Gaussian with log-variance est_uncert; after the W-bipath
            # composition there are two independent terms and the variances add (losses.py:107-119)
            assert self.loss_type in ('L2Loss', 'HuberLoss')
            if est_uncert.shape[1] == 1:
                log_var = est_uncert
            elif est_uncert.shape[1] == 2:
                log_var = torch.logsumexp(est_uncert, 1, keepdim=True)
            else:
                raise ValueError(est_uncert.shape)
            loss = 0.5 * torch.exp(-log_var) * loss + log_var + math.log(2 * math.pi)
        return (loss if mask is None else torch.masked_select(loss, mask)).mean()

    # the reference's entry points, kept for callers that use them directly
    def one_scale(self, est_flow, gt_flow, mask=None):
        return self._one_level(est_flow, None, gt_flow, mask)

    def probabilistic_one_scale(self, est_flow, est_uncert, gt_flow, mask=None):
        return self._one_level(est_flow, est_uncert, gt_flow, mask)

    def forward(self, flow_output, gt_flow, mask=None):
        if not isinstance(flow_output, Sequence):
            flow_output = [flow_output]
        weights = self.level_weights if self.level_weights else [1] * len(flow_output)
        assert len(weights) == len(flow_output)
        total = 0
        for i, (level, weight) in enumerate(zip(flow_output, weights)):    # coarsest level first
            level_mask = mask[i] if mask is not None and isinstance(mask, Sequence) else mask
            flow, uncert = level if isinstance(level, tuple) else (level, None)
            total = total + weight * self._one_level(flow, uncert, gt_flow, level_mask)
        return total


class WBipathLoss(nn.Module):
    def __init__(self, objective='multi_scale_flow_loss', reduction='mean', level_weights=None, loss_type='L1Loss',
                 downsample_gt_flow=True, detach_flow_for_warping=True, visibility_mask=False, alpha_1=0.03,
                 alpha_2=0.5):
        super().__init__()
        if objective != 'multi_scale_flow_loss':
            raise ValueError(objective)
        self.objective = MultiScaleFlowLoss(level_weights=level_weights, loss_type=loss_type,
                                            downsample_gt_flow=downsample_gt_flow, reduction=reduction)
        self.detach_flow_for_warping, self.visibility_mask = detach_flow_for_warping, visibility_mask
        self.alpha_1, self.alpha_2 = alpha_1, alpha_2

    @staticmethod
    def length_sq(x):
        return torch.sum(x ** 2, dim=1)

    @torch.no_grad()
    def get_cyclic_consistency_mask(self, flow_prime_to_source, warped_flow_source_to_target, synthetic_flow):
        """Forward-backward style visibility test (losses.py:232-250): the composition may differ from the synthetic flow
        by at most alpha_1 (|f|^2 + |g|^2 + |w|^2) + alpha_2 (squared pixels)."""
        synthetic_flow = _resize(synthetic_flow, flow_prime_to_source.shape[-2:])
        mag = self.length_sq(flow_prime_to_source) + self.length_sq(warped_flow_source_to_target) + \
            self.length_sq(synthetic_flow)
        err = self.length_sq(flow_prime_to_source + warped_flow_source_to_target - synthetic_flow)
        return ~(err > self.alpha_1 * mag + self.alpha_2)

    def forward(self, estimated_flow_target_prime_to_source, estimated_flow_source_to_target, flow_map, mask_used,
                return_masks=False):
        H, W = flow_map.shape[-2:]
        if not isinstance(estimated_flow_target_prime_to_source, Sequence):
            estimated_flow_target_prime_to_source = [estimated_flow_target_prime_to_source]
        if not isinstance(estimated_flow_source_to_target, Sequence):
            estimated_flow_source_to_target = [estimated_flow_source_to_target]
        composed, masks, cyclic = [], [], []
        for first, second in zip(estimated_flow_target_prime_to_source, estimated_flow_source_to_target):
            probabilistic = isinstance(first, tuple)
            (f_flow, f_unc), (s_flow, s_unc) = (first, second) if probabilistic else ((first, None), (second, None))
            h, w = f_flow.shape[-2:]
            # the flow the second estimate is sampled with: the first one in THIS level's pixels
            wf = f_flow.detach().clone() if self.detach_flow_for_warping else f_flow.clone()
            wf = wf * wf.new_tensor([float(w) / float(W), float(h) / float(H)]).view(1, 2, 1, 1)
            s_warped = warp(s_flow, wf)
            level = f_flow + s_warped
            if probabilistic:
                level = (level, torch.cat((f_unc, warp(s_unc, wf)), 1))
            composed.append(level)
            mask = get_gt_correspondence_mask(wf.detach())
            if mask_used is not None:
                mask = mask & _resize(mask_used.unsqueeze(1).float(), (h, w)).squeeze(1).floor().bool()
            if self.visibility_mask:
                mc = self.get_cyclic_consistency_mask(f_flow.detach(), s_warped.detach(), flow_map)
                mask = mask & mc
                cyclic.append(mc)
            masks.append(mask)
        loss = self.objective(composed, flow_map, mask=masks)
        if return_masks:
            return loss, masks, (cyclic if cyclic else None), composed
        return loss
