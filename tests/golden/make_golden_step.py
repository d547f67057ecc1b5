"""tests/golden/make_golden_step.py -- G13: one full reference `training_step` (DomainAdaptationSegmentationModel) on
a small synthetic batch with augmentation disabled (color_jitter_p=1.0 -> never, blur=False), closed-form weights and
fixed seeds for the three RNGs the step consumes (python `random`, numpy, torch).  Captured: the three losses, the
per-parameter-group gradient norms seen by the optimiser, checksums of the EMA teacher and of the updated student."""
import os
import random

import numpy as np
import torch
import torch.nn as nn

import _ref_import as R
from fill import closed_form_fill, hashed_uniform

HERE = os.path.dirname(os.path.abspath(__file__))
DIMS = [32, 64, 160, 256]
OPT = {"class_path": "torch.optim.AdamW", "init_args": {"lr": 6e-5, "weight_decay": 0.01}}
SCH = {"class_path": "helpers.lr_scheduler.LinearWarmupPolynomialLR",
       "init_args": {"warmup_iters": 1500, "warmup_ratio": 1e-6, "power": 1.0, "max_steps": 40000}}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} kB)")


def make_batch(b, H, W, blk):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    img = lambda k: (hashed_uniform((b, 3, H, W), k) * 4 - 2).astype(np.float32)  # noqa: E731
    trg = img("g13/trg")
    ref = (0.8 * np.roll(trg, (2, -3), (2, 3)) + 0.2 * img("g13/ref")).astype(np.float32)
    lbl = (hashed_uniform((b, H // blk, W // blk), "g13/lbl") * 19).astype(np.int64)
    lbl = np.repeat(np.repeat(lbl, blk, axis=1), blk, axis=2)     # block-constant so that fdist's label down-scaling keeps classes
    lbl[hashed_uniform((b, H, W), "g13/ign") < 0.05] = 255
    return {"image_src": t(img("g13/src")), "semantic_src": t(lbl), "image_trg": t(trg), "image_ref": t(ref)}


class Recorder:
    """stands where Lightning's optimizer wrapper stands: records grad norms per group at step()"""

    def __init__(self, opt):
        self.opt, self.norms = opt, None

    def zero_grad(self):
        self.opt.zero_grad()

    def step(self):
        self.norms = [float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in g["params"] if p.grad is not None)))
                      for g in self.opt.param_groups]
        self.opt.step()


def run_reference(use_hrda, model_type="mit_b0", dims=DIMS):
    sm = R.ref_module("models.segmentation_model")
    mt = R.ref_module("models.backbones.mix_transformer")
    df = R.ref_module("models.heads.daformer")
    sf = R.ref_module("models.heads.segformer")
    ls = R.ref_module("models.losses")
    vg = R.ref_module("models.backbones.vgg")
    ua = R.ref_module("models.heads.uawarpc")
    sched = R.ref_module("helpers.lr_scheduler") if False else None

    class Model(sm.DomainAdaptationSegmentationModel):
        global_step = 0
        logged = {}

        def optimizers(self):
            return self._rec

        def lr_schedulers(self):
            return self._sch

        def manual_backward(self, loss, retain_graph=False):
            loss.backward(retain_graph=retain_graph)

        def log(self, k, v, **kw):
            self.logged[k] = float(v)

        @property
        def device(self):
            return torch.device("cpu")

    model = Model(
        OPT, SCH,
        backbone=mt.MixVisionTransformer(model_type, drop_path_rate=0.0),
        head=df.DAFormerHead(dims, [0, 1, 2, 3], 19, 'multiple_select', dropout_ratio=0.0),
        loss=ls.PixelWeightedCrossEntropyLoss(),
        alignment_backbone=vg.VGG('vgg16', out_indices=[2, 3, 4]),
        alignment_head=ua.UAWarpCHead(in_index=[0, 1], input_transform='multiple_select', estimate_uncertainty=True),
        backbone_lr_factor=0.1, use_refign=True, adapt_to_ref=False, gamma=0.25, enable_fdist=True,
        color_jitter_p=1.0, blur=False, use_hrda=use_hrda, hrda_output_stride=4,
        hrda_scale_attention=sf.SegFormerHead(dims, [0, 1, 2, 3], 19, 'multiple_select', dropout_ratio=0.0))
    closed_form_fill(model)
    model.train()
    opt = torch.optim.AdamW(model.optimizer_parameters(), lr=6e-5, weight_decay=0.01)
    model._rec = Recorder(opt)
    model._sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    return model


def g13():
    torch.set_grad_enabled(True)
    for use_hrda, name, (H, W) in [(False, "step_daformer_96x128", (96, 128)), (True, "step_hrda_128x128", (128, 128))]:
        model = run_reference(use_hrda)
        batch = make_batch(2, H, W, 64 if use_hrda else 32)
        random.seed(77); np.random.seed(77); torch.manual_seed(77)
        model.global_step = 3          # EMA momentum 1 - 1/4
        model.training_step(batch, 0)
        ema = float(sum(p.double().abs().sum() for p in model.ema_parameters()))
        live = float(sum(p.double().abs().sum() for p in model.live_parameters()))
        save(name, losses=np.array([model.logged["train_loss_src"], model.logged["train_loss_featdist_src"],
                                    model.logged["train_loss_uda_trg"]]),
             grad_norms=np.array(model._rec.norms), ema_abs_sum=ema, live_abs_sum=live, size=np.array([H, W]))
        print("   ", model.logged, model._rec.norms)
    torch.set_grad_enabled(False)


def g13_b5():
    """G13-B5: the same step with the bench's networks (MiT-B5, HRDA) on one 512 x 512 (source, target, reference) triple --
    K2/K3-sized tokens per view (256 x 256 views: 4 096 / 1 024 / 256 / 64 tokens per stage, 10 teacher views per image).
    Besides the scalars: strided samples of the decode head's class-weight gradient and of two updated backbone weights."""
    torch.set_grad_enabled(True)
    dims = [64, 128, 320, 512]
    model = run_reference(True, "mit_b5", dims)
    H = W = 512
    batch = make_batch(1, H, W, 64)
    random.seed(78); np.random.seed(78); torch.manual_seed(78)
    model.global_step = 3
    grads = {}
    real = model._rec.step

    def step():
        grads["conv_seg"] = model.head.conv_seg.weight.grad.detach().flatten()[::37].clone().numpy()
        grads["fc1"] = model.backbone.block3[20].mlp.fc1.weight.grad.detach().flatten()[::997].clone().numpy()
        real()
    model._rec.step = step
    model.training_step(batch, 0)
    ema = float(sum(p.double().abs().sum() for p in model.ema_parameters()))
    live = float(sum(p.double().abs().sum() for p in model.live_parameters()))
    save("step_hrda_b5_512x512", losses=np.array([model.logged["train_loss_src"], model.logged["train_loss_featdist_src"],
                                                  model.logged["train_loss_uda_trg"]]),
         grad_norms=np.array(model._rec.norms), ema_abs_sum=ema, live_abs_sum=live, size=np.array([H, W]),
         grad_conv_seg=grads["conv_seg"], grad_fc1=grads["fc1"],
         w_q=model.backbone.block1[0].attn.q.weight.detach().flatten()[::61].numpy(),
         w_fuse=model.head.fuse_layer.bottleneck.conv.weight.detach().flatten()[::9973].numpy())
    print("   ", model.logged, model._rec.norms)
    torch.set_grad_enabled(False)


def g13_k3():
    """G13-K3: BASELINE config 3 -- one reference training_step of the DAFormer model (no HRDA) with MiT-B5 on a b = 2 batch
    of 512 x 1024 crops (cityscapes_acdc/refign_daformer.yaml:11-41 uses these networks; the crop is the K3 size of
    SURVEY section 8), closed-form weights, augmentation off.  Same capture as G13-B5."""
    torch.set_grad_enabled(True)
    dims = [64, 128, 320, 512]
    model = run_reference(False, "mit_b5", dims)
    H, W = 512, 1024
    batch = make_batch(2, H, W, 64)
    random.seed(79); np.random.seed(79); torch.manual_seed(79)
    model.global_step = 3
    grads = {}
    real = model._rec.step

    def step():
        grads["conv_seg"] = model.head.conv_seg.weight.grad.detach().flatten()[::37].clone().numpy()
        grads["fc1"] = model.backbone.block3[20].mlp.fc1.weight.grad.detach().flatten()[::997].clone().numpy()
        real()
    model._rec.step = step
    import time
    t0 = time.perf_counter()
    model.training_step(batch, 0)
    print(f"    reference step: {time.perf_counter() - t0:.1f} s")
    ema = float(sum(p.double().abs().sum() for p in model.ema_parameters()))
    live = float(sum(p.double().abs().sum() for p in model.live_parameters()))
    save("step_daformer_b5_512x1024", losses=np.array([model.logged["train_loss_src"], model.logged["train_loss_featdist_src"],
                                                       model.logged["train_loss_uda_trg"]]),
         grad_norms=np.array(model._rec.norms), ema_abs_sum=ema, live_abs_sum=live, size=np.array([H, W]),
         grad_conv_seg=grads["conv_seg"], grad_fc1=grads["fc1"],
         w_q=model.backbone.block1[0].attn.q.weight.detach().flatten()[::61].numpy(),
         w_fuse=model.head.fuse_layer.bottleneck.conv.weight.detach().flatten()[::9973].numpy())
    print("   ", model.logged, model._rec.norms)
    torch.set_grad_enabled(False)


GROUPS = {"G13": g13, "G13B5": g13_b5, "G13K3": g13_k3}
