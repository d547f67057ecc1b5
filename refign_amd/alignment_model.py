"""Host mirror of models/alignment_model.py (AlignmentModel): constructor keywords of the reference, `forward(images_i,
images_j)` = flow i->j at full resolution and 1 - P_R (alignment_model.py:55-79).  Matcher TRAINING (training_step,
MultiScaleFlowLoss / WBipathLoss) is the "next" row N1 of SURVEY §8f and is not built."""
from typing import Optional

import torch
import torch.nn as nn

from . import align as align_mod


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
        self.selfsupervised_loss, self.unsupervised_loss = selfsupervised_loss, unsupervised_loss   # specs (row N1)
        self.apply_constant_flow_weights = apply_constant_flow_weights
        if pretrained is not None:
            ckpt = torch.load(pretrained, map_location='cpu')
            self.load_state_dict(ckpt.get('state_dict', ckpt), strict=True)

    @torch.no_grad()
    def forward(self, images_i, images_j):
        return align_mod.alignment_forward(self.alignment_backbone, self.alignment_head, images_i, images_j)

    def training_step(self, batch, batch_idx):
        raise NotImplementedError("UAWarpC matcher training is SURVEY §8f N1 (next), not part of the UDA hot path")
