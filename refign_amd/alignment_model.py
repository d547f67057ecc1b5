"""Host mirror of models/alignment_model.py (AlignmentModel): constructor keywords of the reference, `forward(images_i,
images_j)` = flow i->j at full resolution and 1 - P_R (alignment_model.py:55-79), and the matcher TRAINING step
(alignment_model.py:81-146): warp supervision on a synthetically warped `prime` image plus the
W-bipath constraint, three passes through the UAWarpC head per step on a frozen VGG-16 pyramid.

The head passes run under autograd on the hand-written kernels that have a backward (9x9 correlation forward + backward,
csrc/corr.hip; bilinear warp forward + backward, csrc/warp.hip) and library convolutions / BatchNorm for the decoders;
the gradient-free fused kernels of the UDA hot path (correlation + ReLU + L2 norm in one launch, fused uncertainty
front end, implicit-GEMM decoders) are not used here: they have no backward."""
from typing import Optional

import torch
import torch.nn as nn

from . import align as align_mod
from . import config


class AlignmentModel(nn.Module):
    def __init__(self, optimizer_init: dict = None, lr_scheduler_init: dict = None,
                 alignment_backbone: nn.Module = None, alignment_head: nn.Module = None,
                 selfsupervised_loss: nn.Module = None, unsupervised_loss: nn.Module = None, metrics: dict = {},
                 apply_constant_flow_weights: bool = False, pretrained: Optional[str] = None):
        super().__init__()
        self.alignment_backbone = alignment_backbone
        self.alignment_head = alignment_head
        self.alignment_backbone.requires_grad_(False)
        self.optimizer_init, self.lr_scheduler_init = optimizer_init, lr_scheduler_init
        self.selfsupervised_loss, self.unsupervised_loss = selfsupervised_loss, unsupervised_loss
        self.apply_constant_flow_weights = apply_constant_flow_weights
        self.logged = {}
        from .metrics import build_collections
        self.valid_metrics, self.test_metrics = build_collections(metrics, config.instantiate_class)
        if pretrained is not None:
            ckpt = torch.load(pretrained, map_location='cpu')
            self.load_state_dict(ckpt.get('state_dict', ckpt), strict=True)

    def log(self, name, value, **kwargs):
        self.logged[name] = value.detach() if torch.is_tensor(value) else value

    @torch.no_grad()
    def forward(self, images_i, images_j):
        """alignment_model.py:55-79.  In eval mode on a GPU the whole forward (frozen VGG-16, head, up-sampling, confidence: ~100
        launches, no host decision that depends on data) replays from a hipGraph after the first call of a shape
        (refign_amd/graphs.py; K2 was host-bound without it)."""
        if images_i.is_cuda and not self.training:
            g = self.__dict__.get("_fwd_graph")
            if g is None:
                from .graphs import GraphedNoGrad
                g = self.__dict__["_fwd_graph"] = GraphedNoGrad(self._forward_eager, "AlignmentModel.forward")
            return tuple(o.clone() if torch.is_tensor(o) else o for o in g(images_i, images_j))    # the graph's own outputs are overwritten by the next call
        return self._forward_eager(images_i, images_j)

    def _forward_eager(self, images_i, images_j):
        return align_mod.alignment_forward(self.alignment_backbone, self.alignment_head, images_i, images_j)

    # -- evaluation (alignment_model.py:148-190) -----------------------------------------------------------------------
    def _eval_step(self, metrics, batch, src_name):
        """Flow target -> reference at full resolution and its confidence, every metric of this dataset accumulates
        (`src_name`: what the reference reads from trainer.datamodule.idx_to_name[split][dataloader_idx])."""
        images_ref, images_trg = batch['image_ref'], batch['image']
        h, w = images_ref.shape[-2:]
        flow, uncert = self.forward(images_trg, images_ref)
        for k, m in metrics.items():
            if src_name in k:
                m(flow, batch['corr_pts_ref'], batch['corr_pts'], (h, w), uncert)
        return flow, uncert

    def validation_step(self, batch, batch_idx=0, dataloader_idx=0, src_name=""):
        return self._eval_step(self.valid_metrics, batch, src_name)

    def test_step(self, batch, batch_idx=0, dataloader_idx=0, src_name=""):
        return self._eval_step(self.test_metrics, batch, src_name)

    def _epoch_end(self, metrics):
        out = metrics.compute()
        metrics.reset()
        for k, v in out.items():
            self.log(k, v)
        return out

    def validation_epoch_end(self, outs=None):
        return self._epoch_end(self.valid_metrics)

    def test_epoch_end(self, outs=None):
        return self._epoch_end(self.test_metrics)

    def __getstate__(self):
        d = self.__dict__.copy()
        d.pop("_fwd_graph", None)                         # copies and pickles capture their own
        return d

    def _apply(self, fn, *a, **k):
        self.__dict__.pop("_fwd_graph", None)             # cached derived tensors move with the parameters
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.__dict__.pop("_fwd_graph", None)
        return super().load_state_dict(*a, **k)

    def train(self, mode=True):
        """alignment_model.py:233-238: the frozen backbone's norm layers never leave eval mode."""
        self.__dict__.pop("_fwd_graph", None)
        super().train(mode)
        for m in self.alignment_backbone.modules():
            if isinstance(m, nn.modules.batchnorm._BatchNorm):
                m.eval()
        return self

    def configure_optimizers(self):
        """alignment_model.py:192-198: optimiser over the trainable (head) parameters, per-step scheduler."""
        optimizer = config.instantiate_class([p for p in self.parameters() if p.requires_grad], self.optimizer_init)
        scheduler = config.instantiate_class(optimizer, self.lr_scheduler_init)
        return [optimizer], [{'scheduler': scheduler, 'interval': 'step'}]

    @staticmethod
    @torch.no_grad()
    def weights_selfsupervised_and_unsupervised(loss_ss, loss_un, weight_ss=1.0, weight_un=1.0,
                                                apply_constant_weights=False):
        """alignment_model.py:217-231: scale the smaller of the two losses up to the larger one (factor <= 100)."""
        if apply_constant_weights:
            return weight_ss, weight_un
        ratio = weight_ss / weight_un
        if loss_un > loss_ss:
            return torch.clamp(loss_un / loss_ss.clamp(min=1e-8) * ratio, max=100).item(), 1.0
        return 1.0, torch.clamp(loss_ss / loss_un.clamp(min=1e-8) / ratio, max=100).item()

    def _pyramids(self, images, b, n):
        """Frozen VGG-16 features of n stacked image sets at the input resolution (levels -3, -2) and at 256x256 (levels
        -2, -1), split back per set (alignment_model.py:88-104)."""
        images_256 = nn.functional.interpolate(images, size=(256, 256), mode='area')
        with torch.no_grad():
            full = self.alignment_backbone(images, extract_only_indices=[-3, -2])
            small = self.alignment_backbone(images_256, extract_only_indices=[-2, -1])
        return list(zip(*[torch.split(f, [b] * n) for f in full])), list(zip(*[torch.split(f, [b] * n) for f in small]))

    def training_step(self, batch, batch_idx=0):
        """alignment_model.py:81-146.  batch: image_ref, image_trg, image_prime (a synthetic warp of ref or trg, which one
        per sample in prime_trg_idx), flow_prime (the synthetic flow, b x 2 x H x W), mask_prime (b x H x W)."""
        images_ref, images_trg, images_prime = batch['image_ref'], batch['image_trg'], batch['image_prime']
        flow_prime, mask_prime = batch['flow_prime'], batch['mask_prime']
        b, _, h, w = images_trg.shape
        (pyr_ref, pyr_trg, pyr_prime), (pyr_ref_256, pyr_trg_256, pyr_prime_256) = \
            self._pyramids(torch.cat([images_ref, images_trg, images_prime]), b, 3)
        with torch.no_grad():
            # i = the image `prime` was made from, j = the other one of the pair, sample by sample
            idx = [int(v) for v in batch['prime_trg_idx']]

            def pick(pair, which):
                return [torch.stack([pair[(k if which == 0 else 1 - k)][lvl][s] for s, k in enumerate(idx)])
                        for lvl in range(len(pair[0]))]
            pyr_i, pyr_j = pick((pyr_ref, pyr_trg), 0), pick((pyr_ref, pyr_trg), 1)
            pyr_i_256, pyr_j_256 = pick((pyr_ref_256, pyr_trg_256), 0), pick((pyr_ref_256, pyr_trg_256), 1)
        head = self.alignment_head
        prime_i_flow = head(pyr_prime, pyr_i, pyr_prime_256, pyr_i_256, (h, w))        # warp supervision
        prime_j_flow = head(pyr_prime, pyr_j, pyr_prime_256, pyr_j_256, (h, w))        # W-bipath, first leg
        j_i_flow = head(pyr_j, pyr_i, pyr_j_256, pyr_i_256, (h, w))                    # second leg
        ss_loss = self.selfsupervised_loss(prime_i_flow, flow_prime, mask=mask_prime)
        us_loss = self.unsupervised_loss(prime_j_flow, j_i_flow, flow_prime, mask_used=mask_prime)
        # (sic) the reference passes apply_constant_flow_weights in the position of weight_ss (alignment_model.py:140-142):
        # with the flag False the ratio of the two weights is 0, i.e. the W-bipath loss always gets the weight cap (100)
        # when it is the smaller one and the warp-supervision loss gets weight 0 when it is the smaller one -- kept,
        # a trained reference checkpoint saw exactly this objective
        weight_ss, weight_us = self.weights_selfsupervised_and_unsupervised(ss_loss, us_loss,
                                                                            self.apply_constant_flow_weights)
        loss = weight_ss * ss_loss + weight_us * us_loss
        self.log("train_matching_loss", loss, batch_size=b)
        self.log("train_ss_loss", ss_loss)
        self.log("train_us_loss", us_loss)
        return loss
