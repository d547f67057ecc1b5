"""The Refign UDA training step on MI355X.

Host mirror of models/segmentation_model.py (DomainAdaptationSegmentationModel): same constructor keywords (= the
`model.init_args` of the reference's YAML configs), same `training_step` semantics (source CE, ImageNet feature
distance, EMA teacher, align, refine, DACS class-mix, mixed CE, three backward passes, one optimiser step), same
`forward` / `whole_inference` / `slide_inference`, `configure_optimizers`, `align`, `refine`, `eta`,
`update_momentum_encoder`, `train`.  It is a plain nn.Module: PyTorch-Lightning is not part of the product (and not
installed); refign_amd/trainer.py drives it one process per GPU.

What is done differently, for the hardware:
  * align / refine / the logits warp run in the HIP kernels of csrc/ (refign_amd.align / .refine / .matching);
  * the EMA update is one multi-tensor op over ~1090 tensors instead of ~1090 tiny launches (a25);
  * gradients of the three backward passes accumulate in ONE flat fp32 buffer and are all-reduced ONCE per step over
    RCCL (the reference's DDP all-reduces after each of the three manual_backward calls; the mean of sums is the same
    up to summation order) -- see refign_amd/trainer.py;
  * the quirks the reference's numbers depend on are kept: teacher networks run in train mode (stochastic dropout /
    drop-path and batch-statistics BatchNorm), refine() output is not renormalised (D8), the pseudo-label
    weight is batch-global, get_class_masks draws classes from the whole batch.
"""
import contextlib
import copy
import math
import os
import random
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import align as align_mod
from . import refine as refine_mod
from ._tensor import const_tensor, upload_async
from . import dacs as _dacs
from . import f8 as _f8
from .graphs import GraphedNoGrad, GraphedSegment, GraphedSplitStep, GraphedStep
from .params import ema_update
from .config import instantiate_class
from . import seg as _seg
from .seg import DeviceBox, defer_logits, draw_crop_offsets, hrda_backbone, hrda_head, predraw_crop, push_device_crop

IMNET_MEAN = (0.485, 0.456, 0.406)
IMNET_STD = (0.229, 0.224, 0.225)


def crop(img, crop_bbox):
    """helpers/utils.py:44-56."""
    if isinstance(crop_bbox, DeviceBox):                    # crop offsets as device data (graph replay)
        return crop_bbox.crop(img)
    y1, y2, x1, x2 = crop_bbox
    if img.dim() == 4:
        return img[:, :, y1:y2, x1:x2]
    if img.dim() == 3:
        return img[:, y1:y2, x1:x2]
    if img.dim() == 2:
        return img[y1:y2, x1:x2]
    raise NotImplementedError(img.dim())


# ---------------------------------------------------------------------------------------------------------------------
# DACS strong augmentation (helpers/dacs_transforms.py).  The class-mix is exact; colour jitter / blur are random
# augmentations (kornia 0.5.8 in the reference, not installed): restated with torch ops, parity unpinned.
# ---------------------------------------------------------------------------------------------------------------------
def get_class_masks(labels, classes=None):
    """dacs_transforms.py:81-92.  `labels`: (b,1,H,W).  NB the reference draws the candidate classes from
    torch.unique over the WHOLE batch for every sample ("this seems to be a bug, we keep it for consistency") -- so
    the set is computed once here; `classes` lets the caller compute it where the host synchronisation of
    torch.unique is free (start of the step, see training_step).  The random draws are the reference's."""
    if classes is None:
        classes = torch.unique(labels)
    n = classes.shape[0]
    masks = []
    for label in labels:
        choice = np.random.choice(n, int((n + n % 2) / 2), replace=False)
        chosen = classes[upload_async(choice, torch.long, classes.device)]
        masks.append((label.unsqueeze(0) == chosen.view(-1, 1, 1, 1)).sum(0, keepdim=False).unsqueeze(0))
    return masks


def one_mix(mask, data=None, target=None):
    """dacs_transforms.py:102-112: mask selects element 0 (source), 1 - mask element 1 (target)."""
    if mask is None:
        return data, target
    if data is not None:
        m, _ = torch.broadcast_tensors(mask[0], data[0])
        data = (m * data[0] + (1 - m) * data[1]).unsqueeze(0)
    if target is not None:
        m, _ = torch.broadcast_tensors(mask[0], target[0])
        target = (m * target[0] + (1 - m) * target[1]).unsqueeze(0)
    return data, target


def _color_jitter(img01, s):
    """brightness / contrast / saturation / hue jitter of strength s on a [0,1] RGB batch, random order.
    The parameters are drawn from torch's CPU generator, like kornia's ColorJitter in the reference
    (helpers/dacs_transforms.py:49-53) -- NOT from python's `random`, whose stream the reference consumes only for the
    HRDA crop offsets, the adapt_to_ref coin and the DACS switches: a seeded run keeps making the reference's decisions
    on the steps where the jitter fires.  Brightness is additive (kornia 0.5.8), the other three multiplicative."""
    ops = torch.randperm(4).tolist()
    x = img01
    for op in ops:
        u = float(torch.rand(()))
        f = max(0.0, 1 - s) + u * (1 + s - max(0.0, 1 - s))
        if op == 0:
            x = x + (f - 1.0)
        elif op == 1:
            mean = x.mean(dim=(1, 2, 3), keepdim=True)
            x = (x - mean) * f + mean
        elif op == 2:
            gray = (0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3])
            x = (x - gray) * f + gray
        else:
            h = (2.0 * float(torch.rand(())) - 1.0) * s * 2 * math.pi      # rotate chroma in YIQ space
            c, sn = math.cos(h), math.sin(h)
            yiq = np.array([[0.299, 0.587, 0.114], [0.596, -0.274, -0.322], [0.211, -0.523, 0.312]])
            rot = np.array([[1, 0, 0], [0, c, -sn], [0, sn, c]])
            m = (np.linalg.inv(yiq) @ rot @ yiq).tolist()      # 3x3 on the host: applied with scalar multipliers
            r, g, b = x[:, 0:1], x[:, 1:2], x[:, 2:3]          # (no per-step host-to-device copy, no device inverse)
            x = torch.cat([m[i][0] * r + m[i][1] * g + m[i][2] * b for i in range(3)], 1)
        x = x.clamp(0, 1)
    return x


def _gaussian_blur(x, ksize, sigma):
    def k1d(k):
        r = torch.arange(k, dtype=x.dtype, device=x.device) - (k - 1) / 2
        g = torch.exp(-r ** 2 / (2 * sigma ** 2))
        return g / g.sum()
    ky, kx = k1d(ksize[0]), k1d(ksize[1])
    c = x.shape[1]
    x = F.conv2d(F.pad(x, (0, 0, ksize[0] // 2, ksize[0] // 2), mode='reflect'),
                 ky.view(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)
    return F.conv2d(F.pad(x, (ksize[1] // 2, ksize[1] // 2, 0, 0), mode='reflect'),
                    kx.view(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)


def strong_transform(param, data=None, target=None):
    """dacs_transforms.py:14-24: class-mix, then colour jitter (if param['color_jitter'] > p), then blur (> 0.5)."""
    assert data is not None or target is not None
    data, target = one_mix(mask=param['mix'], data=data, target=target)
    if data is not None and data.shape[1] == 3:
        mean = const_tensor(IMNET_MEAN, data).view(1, 3, 1, 1)
        std = const_tensor(IMNET_STD, data).view(1, 3, 1, 1)
        if param['color_jitter'] > param['color_jitter_p']:
            data = (_color_jitter(data * std + mean, param['color_jitter_s']) - mean) / std
        if param['blur'] > 0.5:
            sigma = np.random.uniform(0.15, 1.15)
            ks = tuple(int(np.floor(np.ceil(0.1 * d) - 0.5 + np.ceil(0.1 * d) % 2)) for d in data.shape[2:])
            data = _gaussian_blur(data, ks, sigma)
    return data, target


# ---------------------------------------------------------------------------------------------------------------------
def _upsample_logits(logits, size):
    """F.interpolate(logits, size, mode='bilinear', align_corners=False) of class logits to the image size
    (segmentation_model.py:163, :169, :206, :220, :230, :239).  The decode heads hand over channels-last logits; up-sampled
    in that layout, the 19 x H x W result is channels-last too and everything after it wants NCHW (log_softmax / the
    refine kernels copy 315 MB per image pair at 1080x1920, 0.95 ms each).  Re-laying-out the LOW-resolution logits first
    (20 MB) makes the up-sampled tensor NCHW-contiguous from the start: same arithmetic, no copy."""
    if logits.is_cuda:
        logits = logits.contiguous()
    return F.interpolate(logits, size, mode='bilinear', align_corners=False)


def _logits_for_loss(model, logits, size):
    """_upsample_logits for logits that only `model.loss` consumes: deferred when that loss is the fused one, so that
    up-sampling + cross-entropy + backward run as one kernel (seg.DeferredUpsample / csrc/loss.hip); the up-sampled tensor
    where that does not apply."""
    return defer_logits(logits, size, _seg.fused_ce_consumer(model))


_ALIGN_PREFETCH = True
_MERGE_FD_BACKWARD = True
_EARLY_MIXED_FWD = True
_SRC_BWD_AFTER_TEACHER = True
_SIDE_PRIORITY = 0          # priority of the teacher branch's stream (0: probed for concurrency, -1: rounds 2-5's high-priority stream)


class DomainAdaptationSegmentationModel(nn.Module):
    """models/segmentation_model.py:25-701.  Constructor keywords are the reference's."""

    def __init__(self,
                 optimizer_init: dict,
                 lr_scheduler_init: dict,
                 backbone: nn.Module,
                 head: nn.Module,
                 loss: nn.Module,
                 alignment_backbone: Optional[nn.Module] = None,
                 alignment_head: Optional[nn.Module] = None,
                 metrics: dict = {},
                 backbone_lr_factor: float = 1.0,
                 use_refign: bool = False,
                 use_align: bool = True,
                 gamma: float = 0.25,
                 adapt_to_ref: bool = False,
                 disable_M: bool = False,
                 disable_P: bool = False,
                 ema_momentum: float = 0.999,
                 pseudo_label_threshold: float = 0.968,
                 psweight_ignore_top: int = 0,
                 psweight_ignore_bottom: int = 0,
                 enable_fdist: bool = True,
                 fdist_lambda: float = 0.005,
                 fdist_classes: list = [6, 7, 11, 12, 13, 14, 15, 16, 17, 18],
                 fdist_scale_min_ratio: float = 0.75,
                 color_jitter_s: float = 0.2,
                 color_jitter_p: float = 0.2,
                 blur: bool = True,
                 use_hrda: bool = False,
                 hrda_output_stride: int = 4,
                 hrda_scale_attention: Optional[nn.Module] = None,
                 hr_loss_weight: float = 0.1,
                 use_slide_inference: bool = False,
                 inference_batched_slide: bool = True,
                 inference_crop_size: list = [1080, 1080],
                 inference_stride: list = [420, 420],
                 pretrained: Optional[str] = None):
        super().__init__()
        self.backbone, self.head = backbone, head
        self.hrda_scale_attention = hrda_scale_attention if use_hrda else None
        self.alignment_backbone, self.alignment_head = alignment_backbone, alignment_head
        for m in filter(None, [self.alignment_backbone, self.alignment_head]):
            m.requires_grad_(False)
        # EMA teacher and frozen ImageNet encoder are deep copies taken BEFORE the HRDA wrapping (:77-87)
        self.m_backbone = copy.deepcopy(self.backbone)
        self.m_head = copy.deepcopy(self.head)
        self.m_hrda_scale_attention = copy.deepcopy(self.hrda_scale_attention)
        for p in self.ema_parameters():
            p.requires_grad = False
        self.enable_fdist = enable_fdist
        if enable_fdist:
            self.imnet_backbone = copy.deepcopy(self.backbone)
            self.imnet_backbone.requires_grad_(False)
        self.loss = loss
        # up-sampling + cross-entropy + backward of the student passes as one kernel (seg.DeferredUpsample)
        _seg.mark_fused_ce_consumer(self, loss)
        _seg.mark_fused_ce_consumer(self.head, loss)
        self.metrics_cfg = metrics
        from .metrics import build_collections
        self.valid_metrics, self.test_metrics = build_collections(metrics, instantiate_class)
        self.optimizer_init, self.lr_scheduler_init = optimizer_init, lr_scheduler_init
        self.backbone_lr_factor = backbone_lr_factor
        self.use_refign, self.use_align, self.gamma = use_refign, use_align, gamma
        self.adapt_to_ref, self.disable_M, self.disable_P = adapt_to_ref, disable_M, disable_P
        self.ema_momentum = ema_momentum
        self.pseudo_label_threshold = pseudo_label_threshold
        self.psweight_ignore_top, self.psweight_ignore_bottom = psweight_ignore_top, psweight_ignore_bottom
        self.fdist_lambda, self.fdist_classes = fdist_lambda, fdist_classes
        self.fdist_scale_min_ratio = fdist_scale_min_ratio
        self.color_jitter_s, self.color_jitter_p, self.blur = color_jitter_s, color_jitter_p, blur
        self.use_hrda = use_hrda
        if use_hrda:
            os_ = hrda_output_stride
            self.backbone.forward = hrda_backbone(self.backbone, os_)(self.backbone.forward)
            self.head.forward = hrda_head(self.head, self.hrda_scale_attention, os_)(self.head.forward)
            self.m_backbone.forward = hrda_backbone(self.m_backbone, os_, is_teacher=True)(self.m_backbone.forward)
            self.m_head.forward = hrda_head(self.m_head, self.m_hrda_scale_attention, os_,
                                            is_teacher=True)(self.m_head.forward)
        self.hr_loss_weight = hr_loss_weight
        self.hrda_output_stride = hrda_output_stride
        self.use_slide_inference = use_slide_inference
        self.inference_batched_slide = inference_batched_slide
        self.inference_crop_size, self.inference_stride = inference_crop_size, inference_stride
        self.automatic_optimization = False
        # trainer-provided state
        self.global_step = 0
        self._optimizer = None
        self._scheduler = None
        self._backward = None           # set by the trainer: callable(loss, retain_graph)
        self.logged = {}
        # hipGraph replay of the gradient-free halves (eager until warmed up; eager for good if capture fails)
        self._graphs = {"teacher_backbone": GraphedNoGrad(self._teacher_backbone, "teacher backbone"),
                        "align_refine": GraphedNoGrad(self._align_refine, "align + refine"),
                        # the same in two pieces: the image-only part can be computed a step ahead (prefetch_align_flow)
                        "align_flow": GraphedNoGrad(self._align_flow, "align (images -> flow)"),
                        "tail_refine": GraphedNoGrad(self._tail_refine, "warp + refine")}
        if self.enable_fdist:
            self._graphs["imnet_features"] = GraphedNoGrad(self._imnet_features, "ImageNet features")
        # forward + backward of the student passes (single process; eager under DDP: SyncBatchNorm collectives)
        shared = {}
        self._graphs["source_pass"] = GraphedSplitStep(self._source_fwd, self._source_bwd, "student source pass", shared=shared,
                                                       forward_state=self._student_forward_state, agree=self._any_rank)
        # the mixed pass may run NEXT TO the tail of the source pass (see _training_step_graphed): a memory pool of its own,
        # and its captured backward kernels accumulate into the second flat gradient buffer
        self._graphs["mixed_pass"] = GraphedSplitStep(self._mixed_fwd, self._mixed_bwd, "student mixed pass", shared=None,
                                                      forward_state=self._student_forward_state, agree=self._any_rank,
                                                      capture_context=self._mixed_capture_context,
                                                      after_capture=self._mixed_captured_reduce,
                                                      on_replay=self._mixed_replayed_reduce)
        # data parallelism through torch.distributed (RFN_DDP_MODE=torch): the passes stay eager around the decode heads' statistics
        # exchanges, their collective-free backbones replay from graphs (graphs.GraphedSegment)
        self._graphs["source_backbone"] = GraphedSegment(self._backbone_fn, "student backbone (source pass)")
        self._graphs["mixed_backbone"] = GraphedSegment(self._backbone_fn, "student backbone (mixed pass)")
        self.teacher_f8 = _f8.ENV_DEFAULT                # K5: EMA-teacher backbone in fp8 (no reference analogue)
        self.load_weights(pretrained)

    def _student_forward_state(self):
        """What a student forward updates in place besides computing its outputs: the BatchNorm running statistics and batch
        counters of the decode heads (graphs.GraphedSplitStep keeps them across the eager re-run after a failed capture)."""
        return [b for m in (self.backbone, self.head, self.hrda_scale_attention) if m is not None
                for n, b in m.named_buffers() if n.endswith(("running_mean", "running_var", "num_batches_tracked"))]

    @staticmethod
    def _any_rank(flag):
        """`flag` on any rank of the data-parallel group (a world of one: the flag itself)."""
        from .bn import data_parallel
        if not data_parallel():
            return flag
        import torch.distributed as dist
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=torch.cuda.current_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))

    # -- trainer hooks (what Lightning provides in the reference) ---------------------------------------------------
    @property
    def device(self):
        return next(self.parameters()).device

    def optimizers(self):
        return self._optimizer

    def lr_schedulers(self):
        return self._scheduler

    def manual_backward(self, loss, retain_graph=False, last=False):
        """`last`: the final backward pass of the step -- the trainer may start reducing gradients that are complete
        while it is still running (refign_amd/trainer.py)."""
        if self._backward is not None:
            self._backward(loss, retain_graph, last)
        else:
            loss.backward(retain_graph=retain_graph)

    def log(self, name, value, **kw):
        self.logged[name] = value.detach() if torch.is_tensor(value) else value

    # -- the step ------------------------------------------------------------------------------------------------
    def training_step(self, batch, batch_idx):
        """segmentation_model.py:146-253."""
        opt, sch = self.optimizers(), self.lr_schedulers()
        opt.zero_grad()
        images_src, gt_src = batch['image_src'], batch['semantic_src']
        # class set of the source labels for the DACS class-mix (dacs_transforms.py:84): torch.unique synchronises
        # with the device, so it is taken HERE, with nothing of this step queued yet, not in the middle of the step
        # where it would drain the launch queue (same values: gt_src does not change during the step)
        nb_trg = batch['image_trg'].shape[0]
        src_classes = self._take_class_prefetch(gt_src, nb_trg)
        if src_classes is None:
            src_classes = torch.unique(gt_src[:nb_trg] if gt_src.shape[0] > nb_trg else gt_src)
        self._prefetch_classes(batch.get("semantic_src_next"), nb_trg)
        self.update_momentum_encoder()

        # The teacher branch (teacher forward on target + reference, align, refine) depends on nothing the student does
        # in this step, only on the EMA update above: start it NOW on a side stream so that its kernels fill the gaps
        # of the launch-bound student forward/backward (many small kernels, each well below 256 workgroups).
        early = early_imnet = None
        if self._student_graphs(images_src):
            return self._training_step_graphed(batch, images_src, gt_src, src_classes, opt, sch)
        if self._overlap_teacher(images_src):
            if self.enable_fdist:                                        # needed first (after the source backward)
                early_imnet = self._start_imnet_features(images_src)
            early = self._start_target_branch(batch, images_src)

        # SOURCE (:156-179)
        feats_src = self.backbone(images_src)
        logits_src = self.head(feats_src)
        if self.use_hrda:
            feats_src = feats_src[0]                                     # low-resolution features
            logits_src, hr_logits_src, crop_box_src = logits_src
            logits_src = _logits_for_loss(self, logits_src, images_src.shape[-2:])
            loss_src = (1 - self.hr_loss_weight) * self.loss(logits_src, gt_src) + \
                self.hr_loss_weight * self.loss(hr_logits_src, crop(gt_src, crop_box_src))
        else:
            logits_src = _logits_for_loss(self, logits_src, images_src.shape[-2:])
            loss_src = self.loss(logits_src, gt_src)
        self.log("train_loss_src", loss_src)
        merged = self.enable_fdist and _MERGE_FD_BACKWARD    # one backward pass of loss_src + loss_fd (see _source_bwd)
        if not merged:
            self.manual_backward(loss_src, retain_graph=self.enable_fdist)
            del loss_src
        del logits_src

        # ImageNet feature distance (:181-189)
        if self.enable_fdist:
            if early_imnet is not None:
                early_imnet[1].wait()                                    # current stream waits for the side stream
                for t in early_imnet[0]:
                    t.record_stream(torch.cuda.current_stream())
            loss_fd = self.calc_feat_dist(images_src, gt_src, feats_src,
                                          feat_imnet=None if early_imnet is None else early_imnet[0])
            self.log("train_loss_featdist_src", loss_fd)
            if merged:
                self.manual_backward(loss_src + loss_fd)
                del loss_src
            else:
                self.manual_backward(loss_fd)
            del loss_fd
        del feats_src

        # TARGET: teacher, align, refine, DACS mix (:194-224)
        with torch.no_grad():
            if early is None:
                images_trg, m_probs_trg = self._target_branch(batch)
            else:
                images_trg, m_probs_trg = early
                cur = torch.cuda.current_stream()
                cur.wait_stream(self._side_stream)
                m_probs_trg.record_stream(cur)
            mixed_img, mixed_lbl, mixed_weight = self.get_dacs_mix(images_trg, m_probs_trg, images_src, gt_src,
                                                                   src_classes)

        # MIXED (:226-250)
        mixed_pred = self.head(self.backbone(mixed_img))
        if self.use_hrda:
            mixed_pred, hr_mixed_pred, box = mixed_pred
            mixed_pred = _logits_for_loss(self, mixed_pred, mixed_img.shape[-2:])
            mixed_loss = (1 - self.hr_loss_weight) * self.loss(mixed_pred, mixed_lbl, pixel_weight=mixed_weight) + \
                self.hr_loss_weight * self.loss(hr_mixed_pred, crop(mixed_lbl, box),
                                                pixel_weight=crop(mixed_weight, box))
        else:
            mixed_pred = _logits_for_loss(self, mixed_pred, mixed_img.shape[-2:])
            mixed_loss = self.loss(mixed_pred, mixed_lbl, pixel_weight=mixed_weight)
        self.log("train_loss_uda_trg", mixed_loss)
        self.manual_backward(mixed_loss, last=True)
        del mixed_loss, mixed_pred

        opt.step()
        sch.step()
        self.global_step += 1

    # -- the student passes as replayable units (refign_amd/graphs.py: GraphedStep) ----------------------------------
    def _student_graphs(self, x):
        """Graph replay of the student passes: training on a GPU (see GraphedStep.usable).  HRDA and, since the end of round 4,
        the single-scale DAFormer / SegFormer configurations too (K3: the eager passes kept the host busy for most of a 115 ms
        step, and the line moved with the box's other tenants: profiles/r04_k3_host_sensitivity.txt)."""
        return self.training and torch.is_grad_enabled() and (GraphedStep.usable(x) or self._backbone_segments(x))

    def _backbone_segments(self, x):
        """Data parallelism with every exchange through torch.distributed (bn.ddp_mode() == "torch", the N > 1 default): whole-pass
        graphs are off (a captured torch.distributed collective is a cross-stream branch of the graph: no gain, round 4), so the
        passes run eagerly -- except their backbones, which hold ~90 % of the launches and no collective.  The step then takes the
        same route as the graphed one (device-side HRDA crops, teacher branch queued first); RFN_GRAPH_SEGMENTS=0: all eager."""
        if not (x.is_cuda and os.environ.get("RFN_GRAPH_SEGMENTS", "1") != "0"):
            return False
        from . import graphs
        from .bn import data_parallel, ddp_mode
        return graphs.enabled() and data_parallel() and ddp_mode() == "torch"

    def _backbone_fn(self, images, off):
        if self.use_hrda:
            push_device_crop(off, self.hrda_output_stride * 2.0)
        return self.backbone(images)

    def _backbone(self, slot, images, off):
        """the student backbone of a pass: a replayed segment under torch-mode data parallelism, else the call itself"""
        if self._backbone_segments(images) and not torch.cuda.is_current_stream_capturing():
            return self._graphs[slot + "_backbone"](images, off)
        return self._backbone_fn(images, off)

    def _crop_offsets(self, images, slot):
        """Draw the HRDA crop offsets of the next student forward on the host (python `random`, the reference's stream
        and order: hrda.py:22-27), or take the pre-drawn ones, and put them into the device tensor the pass reads."""
        from . import seg
        if not self.use_hrda:                              # no crop, no draw: a constant input of the replayed pass
            bufs = self.__dict__.setdefault("_crop_off", {})
            dev = bufs.get(slot)
            if dev is None or dev.device != images.device:
                dev = bufs[slot] = torch.zeros(2, dtype=torch.long, device=images.device)
            return dev
        H, W = images.shape[-2:]
        size, div = (int(H * 0.5), int(W * 0.5)), self.hrda_output_stride * 2.0
        if seg._PREDRAWN_CROPS:
            key, off = seg._PREDRAWN_CROPS.pop(0)
            assert key == (H, W, size, div), (key, (H, W, size, div))
        else:
            off = draw_crop_offsets(H, W, size, div)
        bufs = self.__dict__.setdefault("_crop_off", {})
        dev = bufs.get(slot)
        if dev is None or dev.device != images.device:
            dev = bufs[slot] = torch.zeros(2, dtype=torch.long, device=images.device)
        dev.copy_(upload_async(list(off), torch.long, images.device), non_blocking=True)
        return dev

    def _source_fwd(self, images_src, off):
        """SOURCE (:156-163), forward half: backbone + decode head up to the low-resolution class logits."""
        if self.use_hrda:
            feats_src = self._backbone("source", images_src, off)
            logits_src, hr_logits_src, crop_box_src = self.head(feats_src)
            return {"feat": feats_src[0], "logits": logits_src, "hr": hr_logits_src, "box": crop_box_src,
                    "size": tuple(images_src.shape[-2:])}
        feats_src = self._backbone("source", images_src, off)     # single scale (:171-173): `off` is unused
        return {"feat": feats_src, "logits": self.head(feats_src), "size": tuple(images_src.shape[-2:])}

    def _source_bwd(self, held, images_src, gt_src, feat_imnet_last=None):
        """SOURCE (:164-189), loss half: cross-entropy + ImageNet feature distance and their backward.
        `feat_imnet_last`: the frozen ImageNet encoder's last-stage feature of `images_src` when it was computed ahead
        of the step (prefetch_imnet_features); else it is computed here.
        The reference runs two backward passes over the one forward graph (manual_backward(loss_src, retain_graph=True), then
        manual_backward(loss_featdist_src), :179-186), both accumulating into `.grad`: that sum of two gradients is the
        gradient of the sum of the two losses, so ONE backward pass of `loss_src + loss_fd` gives every parameter the same
        gradient (up to the rounding order of the additions) and walks the low-resolution MiT-B5 backbone once instead of
        twice.  (uda._MERGE_FD_BACKWARD = False: the two passes; tests/test_step_gpu.py flips it.)"""
        logits_src = _logits_for_loss(self, held["logits"], held["size"])
        if self.use_hrda:
            loss_src = (1 - self.hr_loss_weight) * self.loss(logits_src, gt_src) + \
                self.hr_loss_weight * self.loss(held["hr"], crop(gt_src, held["box"]))
        else:
            loss_src = self.loss(logits_src, gt_src)
        if not self.enable_fdist:
            self.manual_backward(loss_src)
            return (loss_src.detach(),)
        if not _MERGE_FD_BACKWARD:
            self.manual_backward(loss_src, retain_graph=True)
        loss_fd = self.calc_feat_dist(images_src, gt_src, held["feat"],
                                      feat_imnet=None if feat_imnet_last is None else [feat_imnet_last])
        self.manual_backward(loss_src + loss_fd if _MERGE_FD_BACKWARD else loss_fd)
        return loss_src.detach(), loss_fd.detach()

    def _source_pass_device_crop(self, images_src, gt_src, off, feat_imnet_last=None):
        """SOURCE (:156-179) + ImageNet feature distance (:181-189): forward and backward in one call."""
        return self._source_bwd(self._source_fwd(images_src, off), images_src, gt_src, feat_imnet_last)

    @torch.no_grad()
    def prefetch_imnet_features(self, images_src_next, after=None, stream=None):
        """Software pipelining across steps (optional; Trainer.step(batch, next_batch=...)): the ImageNet feature of the
        NEXT step's source images depends on nothing that training changes (frozen encoder, :98-105), so it can be
        computed on the side stream while THIS step's mixed pass runs alone on the main stream, instead of inside the
        next source pass where it competes with the teacher branch.  Same numbers, same work per step.
        `after`: event on the main stream that the side stream waits for (the source pass of this step has consumed the
        previous prefetch's buffers)."""
        if not (self.enable_fdist and images_src_next.is_cuda and self._overlap_teacher(images_src_next)):
            return
        # (the side stream, behind the teacher branch.  A stream of its own for this -- low priority, so that next step's work only
        # fills gaps -- was measured in round 5: 191-195 ms/step against 133, also with GPU_MAX_HW_QUEUES=8: a fourth busy stream
        # costs far more than the priority inversion it removes; profiles/r05_prefetch_stream_ab.txt)
        self._ensure_side_stream(images_src_next.device)
        st = self._side_stream if stream is None else stream
        if after is not None:
            st.wait_event(after)
        with torch.cuda.stream(st):
            with torch.autocast("cuda", dtype=torch.get_autocast_dtype("cuda"), enabled=torch.is_autocast_enabled("cuda")):
                feat = self._imnet_forward(images_src_next)[-1]
            done = st.record_event()
        self._imnet_prefetch = (images_src_next, images_src_next._version, images_src_next.data_ptr(), feat, done)

    # torch.unique synchronises the host with the device: at the start of a step it drains the whole previous step, and the
    # device then idles while the host enqueues the new one (4.6 ms of every 181 ms step, profiles/r03_step_phases*.txt).
    # With the NEXT batch known (Trainer.step(batch, next_batch=...)) the class set is taken a step ahead WITHOUT a
    # synchronisation: a fixed-size histogram (no data-dependent shape), an asynchronous copy to pinned memory and an event
    # that has long fired when the next step asks for it.  Same values as torch.unique (ascending class ids).
    def _prefetch_classes(self, gt_next, nb):
        self._class_prefetch = None
        if gt_next is None or not gt_next.is_cuda or gt_next.dtype != torch.long:
            return
        g = gt_next[:nb] if gt_next.shape[0] > nb else gt_next
        # labels are 0 .. 18 and 255; histc drops values outside [0, 255] (another ignore convention, -1 ...): the bin total
        # then falls short of the pixel count and _take_class_prefetch declines (torch.unique keeps such labels, and the host's
        # random stream depends on the size of the class set)
        hist = torch.histc(g.to(torch.float32), bins=256, min=0, max=255)
        host = getattr(self, "_class_hist_host", None)
        if host is None:
            host = self._class_hist_host = torch.empty(256, dtype=torch.float32).pin_memory()
        host.copy_(hist, non_blocking=True)
        self._class_prefetch = (gt_next, gt_next._version, gt_next.data_ptr(), nb, torch.cuda.current_stream().record_event(),
                                g.numel())

    def _take_class_prefetch(self, gt_src, nb):
        pf, self._class_prefetch = getattr(self, "_class_prefetch", None), None
        if pf is None or pf[0] is not gt_src or pf[1] != gt_src._version or pf[2] != gt_src.data_ptr() or pf[3] != nb:
            return None
        pf[4].synchronize()                                # recorded a whole step ago
        if int(round(float(self._class_hist_host.sum()))) != pf[5]:
            return None                                    # labels outside 0 .. 255: the caller takes torch.unique
        ids = torch.nonzero(self._class_hist_host > 0).flatten()
        return upload_async(ids, torch.long, gt_src.device)

    def _take_imnet_prefetch(self, images_src):
        pf, self._imnet_prefetch = getattr(self, "_imnet_prefetch", None), None
        if pf is None or pf[0] is not images_src or pf[1] != images_src._version or pf[2] != images_src.data_ptr():
            return None
        torch.cuda.current_stream().wait_event(pf[4])
        pf[3].record_stream(torch.cuda.current_stream())
        return pf[3]

    def _mixed_fwd(self, mixed_img, off):
        """MIXED (:226-231), forward half."""
        if self.use_hrda:
            mixed_pred, hr_mixed_pred, box = self.head(self._backbone("mixed", mixed_img, off))
            return {"logits": mixed_pred, "hr": hr_mixed_pred, "box": box, "size": tuple(mixed_img.shape[-2:])}
        return {"logits": self.head(self._backbone("mixed", mixed_img, off)), "size": tuple(mixed_img.shape[-2:])}

    def _mixed_bwd(self, held, mixed_lbl, mixed_weight):
        """MIXED (:232-250), loss half: pixel-weighted cross-entropy against the (refined) pseudo-labels + backward."""
        mixed_pred = _logits_for_loss(self, held["logits"], held["size"])
        if self.use_hrda:
            box = held["box"]
            mixed_loss = (1 - self.hr_loss_weight) * self.loss(mixed_pred, mixed_lbl, pixel_weight=mixed_weight) + \
                self.hr_loss_weight * self.loss(held["hr"], crop(mixed_lbl, box), pixel_weight=crop(mixed_weight, box))
        else:
            mixed_loss = self.loss(mixed_pred, mixed_lbl, pixel_weight=mixed_weight)
        # the step's LAST backward pass (data parallelism: the finished ranges of the gradient buffer are all-reduced from inside
        # it, trainer._backward / FlatGradBuffer.on_ready) -- when it accumulates into the first buffer in stream order, or into
        # the second buffer next to the source pass AND that buffer is reduced on its own (FlatGradBuffer.use_direct: comm2)
        buf = getattr(self, "_grad_buffer", None)
        self.manual_backward(mixed_loss, last=not getattr(self, "_mixed_on_second", False) or
                             (buf is not None and buf._comm2 is not None))
        return (mixed_loss.detach(),)

    def _mixed_pass_device_crop(self, mixed_img, mixed_lbl, mixed_weight, off):
        """MIXED (:226-250), forward and backward in one call."""
        return self._mixed_bwd(self._mixed_fwd(mixed_img, off), mixed_lbl, mixed_weight)

    def _mixed_captured_reduce(self):
        buf = getattr(self, "_grad_buffer", None)
        if buf is None:
            return ()
        r, buf.captured_ranges = buf.captured_ranges, ()
        return r

    def _mixed_replayed_reduce(self, ranges):
        buf = getattr(self, "_grad_buffer", None)
        if buf is not None and ranges:
            buf.replayed(ranges)

    def _mixed_capture_context(self):
        """What the capture of the mixed pass runs inside when the pass is to run NEXT TO the source pass: parameter
        gradients are views of the second flat buffer (trainer.FlatGradBuffer.into_second).  Under data parallelism both
        passes contain the SyncBatchNorm exchanges of the decode head, and two streams must not issue collectives of
        one communicator in a rank-dependent order.  A communicator per pass through torch's process group was built
        and measured in the 1-rank rehearsal (RFN_DDP_REHEARSAL): 237.9 ms/step against 222.5 ms with the graphed passes
        in stream order -- torch runs a collective on a stream of its own, so a captured one is a cross-stream branch
        of the graph, hipGraph replays those with a synchronisation per edge, and two replays running next to each
        other pay for it.  So under data parallelism the passes stay in stream order, UNLESS the exchanges are RCCL
        calls of our own on the capture stream (refign_amd/rccl.py, RFN_DDP_MODE=direct3: plain kernel nodes) -- then the
        mixed pass gets its own communicator (trainer: model._mixed_comm) and runs next to the source pass."""
        import contextlib
        from .bn import data_parallel, direct_comm
        buf = getattr(self, "_grad_buffer", None)
        comm = getattr(self, "_mixed_comm", None)
        if buf is None or (data_parallel() and comm is None) or os.environ.get("RFN_MIXED_CONCURRENT", "1") == "0":
            self._mixed_on_second = False
            return contextlib.nullcontext()
        self._mixed_on_second = True
        if comm is None:
            return buf.into_second()
        stack = contextlib.ExitStack()
        stack.enter_context(buf.into_second())
        stack.enter_context(direct_comm(comm))
        return stack

    def _mixed_stream(self, x):
        """The stream the DACS mix and the mixed pass run on.  Once both student passes replay from graphs and the mixed
        pass accumulates its gradients into the second flat buffer, that is a stream of its own which waits for the
        TEACHER only: the mixed pass does not depend on the source pass (same weights, separate gradient buffer, separate
        memory pool), so it starts when the pseudo-labels are there and runs next to the last ~40 ms of the source pass
        (the feature-distance backward, which otherwise has the device to itself at 10-25 us per latency-bound kernel).
        What the two passes still share are the BatchNorm running statistics of the decode head, updated by both
        forwards: the mixed pass's head forward comes a whole teacher branch plus its own backbone forward (~88 + ~15 ms here;
        the source pass's head forward is over after ~45 ms) after the source pass's,
        which is why this is limited to the Refign configuration (teacher + align + refine before the mix)."""
        if not (self.use_refign and self.use_align):
            return None
        whole = self._graphs["mixed_pass"].captured() and self._graphs["source_pass"].captured()
        # (torch-mode data parallelism: the passes are eager around their exchanges -- torch.distributed keeps the collectives of the
        # two streams in the host's issue order, the same on every rank -- with their backbones replayed from segments)
        segs = self._backbone_segments(x) and self._graphs["mixed_backbone"].captured() and self._graphs["source_backbone"].captured()
        if not (getattr(self, "_mixed_on_second", False) and (whole or segs) and self._overlap_teacher(x)):
            return None
        if getattr(self, "_mix_stream", None) is None or self._mix_stream.device != x.device:
            # default priority: a second high-priority stream lands on the side stream's hardware queue and the step
            # goes from 181 to 246 ms (measured, round 3)
            # (round 5, with GPU_MAX_HW_QUEUES=8 so that it gets a queue of its own: -0.6 ms, profiles/r05_mix_priority_ab.txt; not adopted)
            # (round 6: the stream is probed for real concurrency with the other two -- graphs.concurrent_stream)
            from .graphs import concurrent_stream
            with torch.cuda.device(x.device):
                self._mix_stream, self._mix_stream_probe = concurrent_stream(
                    x.device, [torch.cuda.current_stream(x.device), getattr(self, "_side_stream", None)])
        return self._mix_stream

    def _training_step_graphed(self, batch, images_src, gt_src, src_classes, opt, sch):
        """training_step with the two student passes replayed from hipGraphs.  Same order of host random draws as the
        reference: source crop, adapt_to_ref coin, DACS parameters, mixed crop.
        The source pass is ONE host call, the teacher branch tens of milliseconds of host work (its decode head and the
        glue between its graphs are eager): the replay goes out FIRST, then the teacher branch is enqueued on the side
        stream behind an event taken before the replay (it depends on the EMA update only) -- so the device runs both
        from the start of the step instead of idling on the main stream while the host enqueues the teacher
        (measured in round 2: 65 ms of every 236 ms step had only the side stream busy)."""
        ready = torch.cuda.current_stream().record_event() if self._overlap_teacher(images_src) else None
        off = self._crop_offsets(images_src, "src")
        cur = torch.cuda.current_stream()
        feat_next = self._take_imnet_prefetch(images_src) if self.enable_fdist else None
        src_graph, mix_graph = self._graphs["source_pass"], self._graphs["mixed_pass"]
        # the source pass as two units (graphs.GraphedSplitStep): the event between them is what the mixed pass's forward waits
        # for -- the decode head's BatchNorm running statistics are updated by both forwards, source first (as in the reference)
        src_graph.forward(images_src, off, variant=feat_next is not None)
        src_fwd_done = cur.record_event()
        # The source pass's BACKWARD waits for the teacher branch (round 5).  The teacher's 40-view kernels fill the chip: small
        # student kernels next to them gain little and cost the teacher 10-15 ms, while two student chains next to EACH OTHER
        # overlap well.  So: teacher + the two student forwards first, then the two backwards side by side (teacher done at
        # 73 instead of 88 ms, step 129.5 -> 127.2 ms; "after the teacher's backbone" 129.2: profiles/r05_src_bwd_hold_ab.txt).
        # Host order: the teacher branch is enqueued BEFORE the source backward's replay, which then waits on its stream.
        mix = early = None
        hold = _SRC_BWD_AFTER_TEACHER and ready is not None
        if hold:
            mix = self._mixed_stream(images_src)
            if mix is not None:
                mix.wait_event(ready)                    # gradient buffers zeroed, EMA done, batch resident
            early = self._start_target_branch(batch, images_src, after=ready)
            if mix is not None:                          # (no mix stream yet = the passes are not captured yet: stream order)
                cur.wait_stream(self._side_stream)
        if feat_next is not None:
            feat_next = feat_next.clone()
            prefetch_free = cur.record_event()           # the prefetch buffer is free again from here on
            losses = src_graph.backward(images_src, gt_src, feat_next)
        else:
            losses = src_graph.backward(images_src, gt_src)
            prefetch_free = cur.record_event()           # (the encoder ran inside the pass: its buffers are busy until then)
        if not hold:
            mix = self._mixed_stream(images_src) if ready is not None else None
            if mix is not None:
                mix.wait_event(ready)                        # gradient buffers zeroed, EMA done, batch resident
            early = None if ready is None else self._start_target_branch(batch, images_src, after=ready)
        self.log("train_loss_src", losses[0])
        if self.enable_fdist:
            self.log("train_loss_featdist_src", losses[1])
        if early is None:
            mix = None
        run_on = cur if mix is None else mix
        if mix is not None and getattr(self, "_grad_buffer", None) is not None:
            # data parallelism: the first gradient buffer is final (the mixed pass accumulates into the second one): its
            # all-reduce runs next to the mixed pass (no-op without the two gradient communicators)
            self._grad_buffer.reduce_first_now()
        nb = batch['image_trg'].shape[0]
        src_nb, gt_nb = (images_src[:nb], gt_src[:nb]) if images_src.shape[0] > nb else (images_src, gt_src)
        # The mixed pass's FORWARD does not need the teacher: the mixed image is cut from the source and target images with a
        # mask of the SOURCE labels (dacs_transforms.py:81-112); only the mixed label / weight, i.e. the loss, need the refined
        # pseudo-labels (:541-574).  On its own stream the forward therefore starts as soon as the source forward has updated the
        # BatchNorm statistics and runs next to the teacher branch; loss + backward follow when the pseudo-labels are there.
        # Same host draws in the same order (source crop, coin, DACS parameters, mixed crop), same numbers.
        early_fwd = (_EARLY_MIXED_FWD and mix is not None and self._dacs_kernels_usable(src_nb, early[0], gt_nb))
        if mix is not None:
            self.__dict__["_mixed_concurrent_steps"] = self.__dict__.get("_mixed_concurrent_steps", 0) + 1   # diagnostics
            if early_fwd:
                mix.wait_event(src_fwd_done)
                self.__dict__["_mixed_early_forwards"] = self.__dict__.get("_mixed_early_forwards", 0) + 1   # diagnostics
            else:
                mix.wait_stream(self._side_stream)       # pseudo-labels
        from .bn import direct_comm
        comm = getattr(self, "_mixed_comm", None)
        seg_second = contextlib.nullcontext
        if self._backbone_segments(images_src) and getattr(self, "_grad_buffer", None) is not None \
                and os.environ.get("RFN_MIXED_CONCURRENT", "1") != "0":
            # torch-mode data parallelism (eager passes, backbone segments): the WHOLE mixed pass -- the capture of its backbone
            # segment, every replay's eager head -- accumulates into the second flat gradient buffer, so that it may run next to the
            # source pass (the host enqueues the two backward passes one after the other: `.grad` points where the pass being
            # enqueued wants it; the optimiser's proxy adds the buffers)
            seg_second = self._grad_buffer.into_second
            self._mixed_on_second = True
        with torch.cuda.stream(run_on), seg_second():
            if early_fwd:
                images_trg, m_probs_trg = early
                with torch.no_grad():
                    d = self._dacs_draw(nb, gt_nb, src_classes)
                    mixed_img, _, _ = _dacs.mix(src_nb, images_trg, gt_nb, None, None, d["bits"], d["jitter"], d["sigma"],
                                                part="image")
                off = self._crop_offsets(mixed_img, "mix")
                with (direct_comm(comm) if comm is not None else contextlib.nullcontext()):
                    mix_graph.forward(mixed_img.to(images_src.dtype).contiguous(), off)
                    mix.wait_stream(self._side_stream)   # pseudo-labels
                    m_probs_trg.record_stream(run_on)
                    with torch.no_grad():
                        pseudo_label, pseudo_weight = self._pseudo_labels(m_probs_trg)
                        _, mixed_lbl, mixed_weight = _dacs.mix(None, None, gt_nb, pseudo_label, pseudo_weight, d["bits"],
                                                               d["jitter"], d["sigma"], part="labels")
                    (mixed_loss,) = mix_graph.backward(mixed_lbl.contiguous(), mixed_weight.to(torch.float32).contiguous())
            else:
                with torch.no_grad():
                    if early is None:
                        images_trg, m_probs_trg = self._target_branch(batch)
                    else:
                        images_trg, m_probs_trg = early
                        if mix is None:
                            cur.wait_stream(self._side_stream)
                        m_probs_trg.record_stream(run_on)
                    mixed_img, mixed_lbl, mixed_weight = self.get_dacs_mix(images_trg, m_probs_trg, images_src, gt_src,
                                                                           src_classes)
                off = self._crop_offsets(mixed_img, "mix")
                # one input signature for the replay: the blur of the DACS augmentation runs under autocast and hands back a
                # 16-bit image on the steps where it fires (a second signature = a second capture with its own eager warm-up)
                # (data parallelism with direct RCCL exchanges: the mixed pass ALWAYS exchanges over its own communicator --
                # eager warm-up, capture, replay and the eager fallback of a failed capture alike -- so that a rank whose
                # capture fails still meets its peers on the communicator their graphs were captured with)
                with (direct_comm(comm) if comm is not None else contextlib.nullcontext()):
                    mix_graph.forward(mixed_img.to(images_src.dtype).contiguous(), off)
                    (mixed_loss,) = mix_graph.backward(mixed_lbl.contiguous(), mixed_weight.to(torch.float32).contiguous())
        if mix is not None:
            cur.wait_stream(mix)                         # both passes done before the optimiser merges their gradients
            mixed_loss.record_stream(cur)
        # the next batch's image-only work (ImageNet features, matcher flow) behind the teacher branch on the side stream.
        # (Measured in round 5 and dropped, profiles/r05_flow_placement_ab.txt: the same work behind the source pass on the main
        # stream, and THIS step's flow on the mix stream while it waits for the source forward -- both within noise.)
        self._prefetch_next(batch, prefetch_free, self._side_stream.record_event() if early is not None else None,
                            early is not None, None)
        self.log("train_loss_uda_trg", mixed_loss)
        if src_graph.any_failed() != mix_graph.any_failed():
            # one student pass could not be captured: the other one goes back to eager launches too (graphs.GraphedSplitStep.give_up)
            (mix_graph if src_graph.any_failed() else src_graph).give_up()
        opt.step()
        sch.step()
        self.global_step += 1

    def _prefetch_next(self, batch, prefetch_free, branch_done, with_flow, stream):
        nxt = batch.get("image_src_next")
        if nxt is not None:                              # after the teacher branch the side stream is idle: fill it
            self.prefetch_imnet_features(nxt, after=prefetch_free, stream=stream)
        if batch.get("image_trg_next") is not None and batch.get("image_ref_next") is not None and with_flow \
                and self._next_step_aligns(batch["image_src"]):
            # (after this step's warp + refine, which reads the flow buffers the prefetch re-fills)
            with torch.no_grad():
                self.prefetch_align_flow(batch["image_ref_next"], batch["image_trg_next"], after=branch_done, stream=stream)

    @torch.no_grad()
    def _target_branch(self, batch):
        """(:194-213) which image is adapted to, and the (refined) teacher probabilities for it."""
        if self.adapt_to_ref and random.random() < 0.5:
            adapt_to_ref, images_trg = True, batch['image_ref']
        else:
            adapt_to_ref, images_trg = False, batch['image_trg']
        if self.use_refign and not adapt_to_ref:
            # teacher forward on (target, reference) + align + refine: gradient-free and shape-static, mostly
            # replayed from hipGraphs after the first eager call (refign_amd/graphs.py)
            return images_trg, self._teacher_align_refine(images_trg, batch['image_ref'])
        # (:210-213) the coin fell on the reference image (or Refign is off): the teacher alone, plain softmax.  The backbone
        # replays from a hipGraph of its own input signature (b images instead of 2b), like the aligned branch's
        m_logits_trg = self.m_head(self._graphs["teacher_backbone"](images_trg))
        m_logits_trg = _upsample_logits(m_logits_trg, images_trg.shape[-2:])
        self.__dict__["_adapted_to_ref_steps"] = self.__dict__.get("_adapted_to_ref_steps", 0) + int(adapt_to_ref)   # diagnostics
        return images_trg, F.softmax(m_logits_trg, dim=1)

    def _overlap_teacher(self, x):
        return x.is_cuda

    def _start_target_branch(self, batch, images_src, after=None):
        """_target_branch on the side stream.  The adapt_to_ref coin is the THIRD draw of the python `random` stream
        in a step (after the two HRDA crop offsets of the source forward): those two are drawn here, in order, and
        handed to the source forward (seg.predraw_crop), so a seeded run makes the same decisions as the reference.
        `after`: an event on the main stream the branch waits for instead of everything queued there so far (the
        caller has already drawn the crop and queued the source pass)."""
        self._ensure_side_stream(images_src.device)
        if after is not None:
            self._side_stream.wait_event(after)
        else:
            if self.adapt_to_ref and self.use_hrda and self.training:
                H, W = images_src.shape[-2:]
                predraw_crop(H, W, (int(H * 0.5), int(W * 0.5)), self.hrda_output_stride * 2.0)
            self._side_stream.wait_stream(torch.cuda.current_stream())   # EMA update (and the batch) are ready
        with torch.cuda.stream(self._side_stream):
            return self._target_branch(batch)

    def _ensure_side_stream(self, device):
        if getattr(self, "_side_stream", None) is None or self._side_stream.device != device:
            # Rounds 2-5: a HIGH-priority stream, because the runtime keeps those in a hardware-queue pool of their own (with the
            # default priority the stream could land on the main stream's queue once an RCCL communicator had taken its streams,
            # and two streams on one queue do not overlap: 328 vs 299 ms/step).  Round 6: default priority and a stream PROBED
            # for concurrency with the main stream (graphs.concurrent_stream).  Same step time in the headline configuration and
            # in the data-parallel rehearsals, and the `adapt_to_ref` configuration goes from 157 to 107 ms per step: there the
            # high-priority teacher starved the source forward (it ended WITH the teacher) once the mixed pass had a queue of
            # its own (profiles/r06_stream_priority_ab.txt).
            if _SIDE_PRIORITY == 0:
                from .graphs import concurrent_stream
                with torch.cuda.device(device):
                    self._side_stream, self._side_stream_probe = concurrent_stream(device, [torch.cuda.current_stream(device)])
                return
            self._side_stream = torch.cuda.Stream(device=device, priority=_SIDE_PRIORITY)

    def _teacher_align_refine(self, images_trg, images_ref):
        """segmentation_model.py:201-213: EMA-teacher logits of (target, reference), warp of the reference logits onto
        the target (align), adaptive label correction (refine) -> refined target probabilities.  The teacher backbone
        and align + refine replay from hipGraphs (refign_amd/graphs.py); the decode head in between stays eager: its
        BatchNorms run in train mode (D9), i.e. they are collectives under DDP with sync_batchnorm."""
        b = images_trg.shape[0]
        m_input = torch.cat((images_trg, images_ref))
        m_logits = self.m_head(self._graphs["teacher_backbone"](m_input))
        m_logits = _upsample_logits(m_logits, m_input.shape[-2:])
        m_logits_trg, m_logits_ref = torch.split(m_logits, [b, b], dim=0)
        if self.use_align:
            if self._align_split(images_trg):
                # two pieces, so that the image-only one can come from the previous step (prefetch_align_flow); the same two
                # graphs either way: a run with and one without the prefetch compute the same numbers
                flow = self._take_align_prefetch(images_ref, images_trg)
                if flow is None:
                    flow = self._graphs["align_flow"](images_ref, images_trg)
                return self._graphs["tail_refine"](m_logits_trg.contiguous(), m_logits_ref.contiguous(), *flow)
            return self._graphs["align_refine"](m_logits_trg.contiguous(), m_logits_ref.contiguous(), images_ref,
                                                images_trg)
        return self.refine(m_logits_trg, m_logits_ref, None, None)

    def _align_refine(self, logits_trg, logits_ref, images_ref, images_trg):
        warped, warp_mask, warp_certs = self.align(logits_ref, images_ref, images_trg)
        return self.refine(logits_trg, warped, warp_mask, warp_certs)

    def _align_flow(self, images_ref, images_trg):
        return align_mod.align_flow(self.alignment_backbone, self.alignment_head, images_ref, images_trg)

    def _tail_refine(self, logits_trg, logits_ref, flow_q, logvar_q):
        warped, warp_mask, warp_certs = align_mod.align_from_flow(logits_ref, flow_q, logvar_q)
        return self.refine(logits_trg, warped, warp_mask, warp_certs)

    def _align_split(self, x):
        """align() as flow + (warp, refine): whenever the flow of the next batch may be computed ahead, i.e. the target
        branch runs on the side stream.  With `adapt_to_ref` (refign_hrda_star.yaml:92) the branch aligns on tails only: the
        prefetch then goes by a PEEK at the next step's coin (_next_step_aligns); a flow that was not prefetched is computed
        in the step by the same graph, an unused one is dropped -- the numbers never depend on the prediction.
        (module constant _ALIGN_PREFETCH = False: align() in one piece.)"""
        return _ALIGN_PREFETCH and self.use_refign and self.use_align and x.is_cuda and self._overlap_teacher(x)

    def _next_step_aligns(self, images_src):
        """Will the NEXT training_step take the Refign branch?  Without `adapt_to_ref`: always.  With it, the coin is the next
        step's third draw from python's `random` stream, after the two HRDA crop offsets of its source forward (the order
        the reference draws in: hrda.py:22-27, then segmentation_model.py:195): those draws are made here on a COPY of the
        generator state, which is put back -- a prediction that consumes nothing.  (A loader that draws from `random`
        between the steps only makes it a worse prediction.)"""
        if not self.adapt_to_ref:
            return True
        state = random.getstate()
        try:
            if self.use_hrda:
                H, W = images_src.shape[-2:]
                draw_crop_offsets(H, W, (int(H * 0.5), int(W * 0.5)), self.hrda_output_stride * 2.0)
            return not random.random() < 0.5
        finally:
            random.setstate(state)

    def prefetch_align_flow(self, images_ref_next, images_trg_next, after=None, stream=None):
        """Software pipelining across steps, like prefetch_imnet_features: the matcher (frozen VGG-16 + flow decoders,
        ~16 ms of the teacher branch at 1080 x 1920) sees the two images only, so the flow of the NEXT batch is computed on the
        side stream while this step's mixed pass runs -- the teacher branch, the head of the critical
        path teacher -> mixed pass, gets that much shorter.  Same work per step, same numbers (same two graphs).
        `after`: event the stream waits for (this step's warp + refine has consumed the previous result)."""
        if not self._align_split(images_trg_next):
            return
        # the side stream itself (behind the teacher branch and the ImageNet-feature prefetch): a fourth stream of the
        # process shares a hardware queue with one of the other three and the step goes from 157 to 301 ms (measured)
        self._ensure_side_stream(images_trg_next.device)
        st = self._side_stream if stream is None else stream
        if after is not None:
            st.wait_event(after)
        with torch.cuda.stream(st):
            flow = self._graphs["align_flow"](images_ref_next, images_trg_next)
            done = st.record_event()
        key = tuple((t, t._version, t.data_ptr()) for t in (images_ref_next, images_trg_next))
        self._align_prefetch = (key, flow, done)

    def _take_align_prefetch(self, images_ref, images_trg):
        pf, self._align_prefetch = getattr(self, "_align_prefetch", None), None
        if pf is None:
            return None
        key, flow, done = pf
        for (t, ver, ptr_), cur in zip(key, (images_ref, images_trg)):
            if t is not cur or t._version != ver or t.data_ptr() != ptr_:
                return None
        torch.cuda.current_stream().wait_event(done)       # (the flow tensors are the graph's own output buffers)
        self.__dict__["_align_prefetch_used"] = self.__dict__.get("_align_prefetch_used", 0) + 1      # diagnostics
        return flow

    def _teacher_backbone(self, x):
        # K5 (model.teacher_f8 / bench.py --precision k5): the EMA teacher's MiT blocks on the fp8 matrix-core kernels
        # (refign_amd/f8.py); everything else of the step is unchanged
        with _f8.teacher_f8(self.teacher_f8 and x.is_cuda and torch.is_autocast_enabled("cuda")
                            and torch.get_autocast_dtype("cuda") == torch.bfloat16):
            return self.m_backbone(x)

    def _imnet_features(self, img):
        f = self.imnet_backbone(img)
        return [t.detach() for t in f] if isinstance(f, Sequence) else [f.detach()]

    def _reset_graphs(self):
        for g in getattr(self, "_graphs", {}).values():
            g.reset()

    def _apply(self, fn, *a, **k):
        self._reset_graphs()                         # .to() / .cuda() / .half(): cached copies and buffers move
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._reset_graphs()
        return super().load_state_dict(*a, **k)

    # -- evaluation (:255-281) ---------------------------------------------------------------
    @torch.no_grad()
    def _eval_step(self, metrics, batch, src_name):
        """validation_step / test_step: logits at the label size, every metric of this dataset accumulates.  `src_name`
        is what the reference reads from trainer.datamodule.idx_to_name[split][dataloader_idx]."""
        x, y = batch['image'], batch['semantic']
        y_hat = self.forward(x, out_size=y.shape[-2:])
        for k, m in metrics.items():
            if src_name in k:
                m(y_hat, y)
        return y_hat

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

    # Cityscapes train-id colours (the dataset's public label definition), (R, G, B) per train id
    CITYSCAPES_COLOURS = (
        ("road", (128, 64, 128)), ("sidewalk", (244, 35, 232)), ("building", (70, 70, 70)), ("wall", (102, 102, 156)),
        ("fence", (190, 153, 153)), ("pole", (153, 153, 153)), ("traffic light", (250, 170, 30)),
        ("traffic sign", (220, 220, 0)), ("vegetation", (107, 142, 35)), ("terrain", (152, 251, 152)),
        ("sky", (70, 130, 180)), ("person", (220, 20, 60)), ("rider", (255, 0, 0)), ("car", (0, 0, 142)),
        ("truck", (0, 0, 70)), ("bus", (0, 60, 100)), ("train", (0, 80, 100)), ("motorcycle", (0, 0, 230)),
        ("bicycle", (119, 11, 32)))

    @torch.no_grad()
    def predict_step(self, batch, batch_idx=0, dataloader_idx=0, save_dir=None, orig_size=None):
        """segmentation_model.py:283-302: arg-max label maps of a batch as PNGs -- `<save_dir>/preds/<filename>` holds the
        train ids (8-bit), `<save_dir>/color_preds/<filename>` the same image with the Cityscapes palette.  The reference
        takes `save_dir` from the Lightning checkpoint directory and `orig_size` from the data module."""
        from PIL import Image
        preds = torch.argmax(self.forward(batch['image'], orig_size), dim=1).to(torch.uint8).cpu().numpy()
        if save_dir is not None:
            pal = [v for _, rgb in self.CITYSCAPES_COLOURS for v in rgb]
            pal += [0] * (768 - len(pal))
            for sub in ('preds', 'color_preds'):
                os.makedirs(os.path.join(save_dir, sub), exist_ok=True)
            for arr, name in zip(preds, batch['filename']):
                image = Image.fromarray(arr)
                image.save(os.path.join(save_dir, 'preds', name))
                col = image.convert('P')
                col.putpalette(pal)
                col.save(os.path.join(save_dir, 'color_preds', name))
        return preds

    # -- inference (:304-382) ------------------------------------------------------------------------------------
    def forward(self, x, out_size=None):
        logits = self.slide_inference(x) if self.use_slide_inference else self.whole_inference(x)
        if out_size is not None:
            logits = F.interpolate(logits, size=out_size, mode='bilinear', align_corners=False)
        return logits

    def whole_inference(self, x):
        logits = self.head(self.backbone(x))
        return F.interpolate(logits, x.shape[-2:], mode='bilinear', align_corners=False)

    def slide_inference(self, img):
        hs, ws = self.inference_stride
        hc, wc = self.inference_crop_size
        b, _, H, W = img.shape
        boxes = []
        for iy in range(max(H - hc + hs - 1, 0) // hs + 1):
            for ix in range(max(W - wc + ws - 1, 0) // ws + 1):
                y2, x2 = min(iy * hs + hc, H), min(ix * ws + wc, W)
                boxes.append((max(y2 - hc, 0), y2, max(x2 - wc, 0), x2))
        preds = img.new_zeros((b, self.head.num_classes, H, W))
        count = img.new_zeros((b, 1, H, W))
        if self.inference_batched_slide:
            logits = self.whole_inference(torch.cat([img[:, :, y1:y2, x1:x2] for y1, y2, x1, x2 in boxes], dim=0))
            for i, (y1, y2, x1, x2) in enumerate(boxes):
                preds[:, :, y1:y2, x1:x2] += logits[i * b:(i + 1) * b]
                count[:, :, y1:y2, x1:x2] += 1
        else:
            for y1, y2, x1, x2 in boxes:
                preds[:, :, y1:y2, x1:x2] += self.whole_inference(img[:, :, y1:y2, x1:x2])
                count[:, :, y1:y2, x1:x2] += 1
        assert (count == 0).sum() == 0
        return preds / count

    # -- optimisation (:384-419) ---------------------------------------------------------------------------------
    def optimizer_parameters(self):
        groups = {k: [] for k in ('head_weight', 'head_bias', 'backbone_weight', 'backbone_bias')}
        for name, p in self.named_parameters():
            if not p.requires_grad:
                continue
            where = 'backbone' if name.startswith('backbone') else 'head'
            groups[f"{where}_{'bias' if p.dim() == 1 else 'weight'}"].append(p)      # 1-D: biases and norm params
        lr = self.optimizer_init['init_args']['lr']
        wd = self.optimizer_init['init_args']['weight_decay']
        return [
            {'name': 'head_weight', 'params': groups['head_weight'], 'lr': lr, 'weight_decay': wd},
            {'name': 'head_bias', 'params': groups['head_bias'], 'lr': lr, 'weight_decay': 0},
            {'name': 'backbone_weight', 'params': groups['backbone_weight'], 'lr': self.backbone_lr_factor * lr,
             'weight_decay': wd},
            {'name': 'backbone_bias', 'params': groups['backbone_bias'], 'lr': self.backbone_lr_factor * lr,
             'weight_decay': 0},
        ]

    def configure_optimizers(self):
        optimizer = instantiate_class(self.optimizer_parameters(), self.optimizer_init)
        scheduler = instantiate_class(optimizer, self.lr_scheduler_init)
        return [optimizer], [scheduler]

    def load_weights(self, pretrain_path):
        if pretrain_path is None:
            return
        ckpt = torch.load(pretrain_path, map_location='cpu')
        self.load_state_dict(ckpt.get('state_dict', ckpt), strict=True)

    # -- refign (:438-523) ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def refine(self, logits_trg, logits_ref, warp_mask, certs):
        return refine_mod.refine(logits_trg, logits_ref, warp_mask, certs, gamma=self.gamma, disable_M=self.disable_M,
                                 disable_P=self.disable_P)

    @staticmethod
    @torch.no_grad()
    def eta(logits):
        """normalised entropy (:484-491)"""
        ent = -(F.softmax(logits, dim=1) * F.log_softmax(logits, dim=1)).sum(dim=1)
        return ent / math.log(logits.shape[1])

    @torch.no_grad()
    def align(self, logits_ref, images_ref, images_trg):
        assert self.alignment_head is not None
        return align_mod.align(self.alignment_backbone, self.alignment_head, logits_ref, images_ref, images_trg)

    # -- DACS (:525-582) -----------------------------------------------------------------------------------------
    def _dacs_draw(self, nb, gt_src, src_classes):
        """The host draws of one get_dacs_mix call in the reference's order (:533-540 python `random` coins; then per sample
        dacs_transforms.py: class choice (numpy), jitter (torch CPU generator), blur sigma (numpy)) for the HIP-kernel form of
        the mix.  Nothing here looks at the teacher's output."""
        params = {'mix': None, 'color_jitter': random.uniform(0, 1), 'color_jitter_s': self.color_jitter_s,
                  'color_jitter_p': self.color_jitter_p, 'blur': random.uniform(0, 1) if self.blur else 0}
        classes = torch.unique(gt_src) if src_classes is None else src_classes
        bits = _dacs.draw_class_bits(classes, nb)
        jit, sig = [], []
        for i in range(nb):
            jit.append(_dacs.draw_jitter(self.color_jitter_s) if params['color_jitter'] > params['color_jitter_p']
                       else None)
            sig.append(np.random.uniform(0.15, 1.15) if params['blur'] > 0.5 else None)
        return {"bits": bits, "jitter": jit, "sigma": sig}

    def _dacs_kernels_usable(self, images_src, images_trg, gt_src):
        return os.environ.get("RFN_DACS_KERNEL", "1") != "0" and _dacs.usable(
            images_src, images_trg, gt_src, getattr(getattr(self, "head", None), "num_classes", 19))

    def _pseudo_labels(self, probs_trg):
        """(:541-556) arg-max pseudo-label and ONE confidence weight for the whole batch."""
        pseudo_prob, pseudo_label = torch.max(probs_trg, dim=1)
        # ONE scalar for the whole batch: fraction of confident pixels (:552-556)
        weight = (pseudo_prob >= self.pseudo_label_threshold).sum() / pseudo_label.numel()
        pseudo_weight = torch.full_like(pseudo_prob, 1.0) * weight
        if self.psweight_ignore_top > 0:
            pseudo_weight[:, :self.psweight_ignore_top, :] = 0
        if self.psweight_ignore_bottom > 0:
            pseudo_weight[:, -self.psweight_ignore_bottom:, :] = 0
        return pseudo_label, pseudo_weight

    @torch.no_grad()
    def get_dacs_mix(self, images_trg, probs_trg, images_src, gt_src, src_classes=None):
        nb = images_trg.shape[0]
        if images_src.shape[0] > nb:
            images_src, gt_src = images_src[:nb], gt_src[:nb]
        cls = DomainAdaptationSegmentationModel          # (tests call this with a bare namespace for `self`)
        if cls._dacs_kernels_usable(self, images_src, images_trg, gt_src):
            # N4: the pixel work of the mix / jitter / blur as HIP kernels (refign_amd/dacs.py); the draws are made in
            # the order the per-sample loop further down makes them
            d = cls._dacs_draw(self, nb, gt_src, src_classes)
            pseudo_label, pseudo_weight = cls._pseudo_labels(self, probs_trg)
            return _dacs.mix(images_src, images_trg, gt_src, pseudo_label, pseudo_weight, d["bits"], d["jitter"], d["sigma"])
        params = {'mix': None, 'color_jitter': random.uniform(0, 1), 'color_jitter_s': self.color_jitter_s,
                  'color_jitter_p': self.color_jitter_p, 'blur': random.uniform(0, 1) if self.blur else 0}
        pseudo_label, pseudo_weight = cls._pseudo_labels(self, probs_trg)
        gt_weight = torch.ones_like(pseudo_weight)
        masks = get_class_masks(gt_src.unsqueeze(1), src_classes)
        mixed_img, mixed_lbl = [None] * nb, [None] * nb
        for i in range(nb):
            params['mix'] = masks[i]
            mixed_img[i], mixed_lbl[i] = strong_transform(params, data=torch.stack((images_src[i], images_trg[i])),
                                                          target=torch.stack((gt_src[i], pseudo_label[i])))
            _, w = strong_transform(params, target=torch.stack((gt_weight[i], pseudo_weight[i])))
            pseudo_weight[i] = w[0, 0] if w.dim() == 4 else w[0]
        return torch.cat(mixed_img), torch.cat(mixed_lbl).squeeze(1), pseudo_weight

    # -- feature distance (:584-668) -----------------------------------------------------------------------------
    @torch.no_grad()
    def _imnet_forward(self, img):
        if self.use_hrda:
            img = F.interpolate(img, scale_factor=0.5, mode='bilinear', align_corners=False)
        return self._graphs["imnet_features"](img) if "imnet_features" in self._graphs else self._imnet_features(img)

    def _start_imnet_features(self, images_src):
        """ImageNet features of the source images (frozen encoder, no gradient) on the side stream, ahead of the
        teacher branch; returns (features, event the consumer waits for)."""
        cur = torch.cuda.current_stream()
        self._ensure_side_stream(images_src.device)
        self._side_stream.wait_stream(cur)
        with torch.cuda.stream(self._side_stream):
            feats = self._imnet_forward(images_src)
            ev = torch.cuda.Event()
            ev.record(self._side_stream)
        return feats, ev

    def calc_feat_dist(self, img, gt, feat=None, feat_imnet=None):
        assert self.enable_fdist
        if feat_imnet is None:
            feat_imnet = self._imnet_forward(img)
        if not isinstance(feat, Sequence):
            feat = [feat]
        if self.fdist_classes is not None:
            scale = gt.shape[-1] // feat[-1].shape[-1]
            gt_small = self.downscale_label_ratio(gt.unsqueeze(1), scale, self.fdist_scale_min_ratio,
                                                  self.head.num_classes, 255, out_size=feat[-1].shape[-2:]).long()
            cls = const_tensor(self.fdist_classes, gt, dtype=torch.long)
            mask = torch.any(gt_small[..., None] == cls, -1)
            dist = self.masked_feat_dist(feat[-1], feat_imnet[-1], mask)
        else:
            dist = self.masked_feat_dist(feat[-1], feat_imnet[-1])
        return self.fdist_lambda * dist

    @staticmethod
    def masked_feat_dist(f1, f2, mask=None):
        d = torch.norm(f1 - f2, dim=1, p=2)
        if mask is not None:
            # mean over the masked pixels (reference: torch.mean(d[mask]); NaN for an empty mask either way) as a
            # masked sum: boolean-mask indexing needs the element count on the host, i.e. a device synchronisation in
            # the forward and another one in the backward
            m = mask.squeeze(1)
            return torch.where(m, d, torch.zeros_like(d)).sum() / m.sum().to(d.dtype)
        return torch.mean(d)

    @staticmethod
    def downscale_label_ratio(gt, scale_factor, min_ratio, n_classes, ignore_index=255, out_size=None):
        """(:637-668) majority class per scale x scale window if its share >= min_ratio, else ignore.
        `out_size` (extension): the feature-map size to match when H or W is not a multiple of scale_factor -- e.g.
        1080 rows at stride 64, where the MiT stage-4 map has ceil(H/64) rows.  The reference asserts divisibility
        (:661-667) and trains on 1024^2 crops; for divisible sizes the windows, hence the result, are identical."""
        assert scale_factor > 1
        b, c, H, W = gt.shape
        assert c == 1
        oh, ow = -(-H // scale_factor), -(-W // scale_factor)
        if gt.is_cuda and gt.dtype == torch.long and n_classes <= 32 and gt.is_contiguous() and \
                (out_size is None or tuple(out_size) == (oh, ow)) and \
                (out_size is not None or (H % scale_factor == 0 and W % scale_factor == 0)):
            # one kernel instead of a 330 MB one-hot tensor, a pooling pass and a max (csrc/refine.hip)
            from . import _lib
            from ._tensor import current_stream, on_device, ptr
            out = torch.empty((b, 1, oh, ow), dtype=torch.long, device=gt.device)
            with on_device(gt.device):
                rc = _lib.load_library().rfn_label_majority(ptr(gt), ptr(out), b, H, W, int(scale_factor), int(n_classes),
                                                            int(ignore_index), float(min_ratio), current_stream(gt.device))
            _lib.check(rc, "label_majority")
            return out
        out = gt.clone()
        out[out == ignore_index] = n_classes
        # one-hot directly in contiguous NCHW (the reference's F.one_hot(...).permute(...) is a channels-last int64
        # tensor: 330 MB and a 14 ms pooling kernel at 1080x1920)
        onehot = (out == torch.arange(n_classes + 1, device=gt.device).view(1, -1, 1, 1)).float()
        if out_size is None or (H % scale_factor == 0 and W % scale_factor == 0):
            pooled = F.avg_pool2d(onehot, kernel_size=scale_factor)
        else:
            pooled = F.avg_pool2d(onehot, kernel_size=scale_factor, ceil_mode=True)
            if tuple(pooled.shape[-2:]) != tuple(out_size):
                pooled = F.adaptive_avg_pool2d(onehot, tuple(out_size))
        ratio, out = torch.max(pooled, dim=1, keepdim=True)
        out[out == n_classes] = ignore_index
        out[ratio < min_ratio] = ignore_index
        return out

    # -- EMA (:670-689) --------------------------------------------------------------------------------------------
    def grad_ready_groups(self):
        """Trainable parameters in the order their gradients become final during a backward pass: decode heads + MiT
        stage 4 first (mark `stage4`), then stage 3, stage 2, and stage 1 last -- the order of the flat gradient buffer,
        so that each mark releases one contiguous range."""
        seen, groups = set(), []

        def take(tag, mods):
            ps = [p for m in mods if m is not None for p in m.parameters() if p.requires_grad and id(p) not in seen]
            seen.update(id(p) for p in ps)
            groups.append((tag, ps))

        bb = self.backbone
        stage = lambda s: [getattr(bb, f"patch_embed{s}", None), getattr(bb, f"block{s}", None), getattr(bb, f"norm{s}", None)]  # noqa: E731
        if hasattr(bb, "patch_embed4"):
            take("stage4", [self.head, self.hrda_scale_attention] + stage(4))
            take("stage3", stage(3))
            take("stage2", stage(2))
        take("rest", [self])
        return groups

    def ema_parameters(self):
        for m in filter(None, [self.m_backbone, self.m_head, self.m_hrda_scale_attention]):
            yield from m.parameters()

    def live_parameters(self):
        for m in filter(None, [self.backbone, self.head, self.hrda_scale_attention]):
            yield from m.parameters()

    @torch.no_grad()
    def update_momentum_encoder(self):
        m = min(1.0 - 1 / (float(self.global_step) + 1.0), self.ema_momentum)
        # ONE kernel launch for the ~1090 parameter tensors (csrc/reduce.hip) + the refresh of the teacher's cached 16-bit /
        # packed weight copies from a plan built once (params.py)
        lists = self.__dict__.get("_ema_lists")
        if lists is None:                             # the parameter objects never change: walk the module tree once
            lists = self.__dict__["_ema_lists"] = (list(self.ema_parameters()), list(self.live_parameters()))
        ema_update(lists[0], lists[1], m, plan_key=("ema", id(self)))
        if self.teacher_f8:
            _f8.requantize()                          # K5: e4m3 weight copies follow the refreshed bf16 copies, in place

    def train(self, mode=True):
        """(:691-701) alignment nets and the ImageNet encoder always in eval; the reference's attempt to disable
        dropout/drop-path of the teacher tests the TOP-LEVEL modules only and therefore changes nothing (D7)."""
        super().train(mode=mode)
        self._reset_graphs()                         # folded-BN caches and mode-dependent paths are re-made
        for m in filter(None, [self.alignment_backbone, self.alignment_head]):
            m.eval()
        if self.enable_fdist:
            self.imnet_backbone.eval()
        return self
