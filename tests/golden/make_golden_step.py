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


def run_reference(use_hrda, model_type="mit_b0", dims=DIMS, adapt_to_ref=False, enable_fdist=True):
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
        backbone_lr_factor=0.1, use_refign=True, adapt_to_ref=adapt_to_ref, gamma=0.25, enable_fdist=enable_fdist,
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


def g13_k4(H=1080, W=1920, enable_fdist=False, name="step_hrda_b5_1080x1920", seed=80, blk=120):
    """G13-K4: BASELINE config 4 as a STEP at the size the metric is quoted on -- one reference training_step of the HRDA
    MiT-B5 model (refign_hrda_star.yaml's networks) on ONE 1080 x 1920 (source, target, reference) triple, fp32, closed-form
    weights, augmentation off: the student's 540 x 960 low-resolution view + 540 x 960 detail crop, the teacher's 2 x (1 + 9)
    views with the 3 x 3 slide fusion on non-square crops, align at 1080 x 1920, refine, the DACS mix and the fused CE at
    the full size.  Captured as in G13-B5, plus strided samples of the refined target probabilities (what get_dacs_mix
    receives, segmentation_model.py:214), their arg-max / max, and of the mixed label and weight.
    `enable_fdist=False` at 1080 x 1920: the REFERENCE ITSELF cannot compute the feature distance at this size -- the label
    down-scaling pools 1080 rows by 64 into 16 (segmentation_model.py:655-667) while MiT's stage-4 map has 17, and
    masked_feat_dist fails with an IndexError (:634); it trains on 1024 x 1024 crops (refign_hrda_star.yaml:17-20).  The
    feature distance at the K4 scale is pinned by the second fixture, G13-K4F: the same step WITH it at 1088 x 1920
    (17 x 64 rows, the nearest size the reference accepts)."""
    torch.set_grad_enabled(True)
    dims = [64, 128, 320, 512]
    model = run_reference(True, "mit_b5", dims, enable_fdist=enable_fdist)
    model.logged = {}
    batch = make_batch(1, H, W, blk)
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    model.global_step = 3
    grads, seen = {}, {}
    real, real_mix = model._rec.step, model.get_dacs_mix

    def step():
        grads["conv_seg"] = model.head.conv_seg.weight.grad.detach().flatten()[::37].clone().numpy()
        grads["fc1"] = model.backbone.block3[20].mlp.fc1.weight.grad.detach().flatten()[::997].clone().numpy()
        real()

    def mix(images_trg, probs_trg, images_src, gt_src):
        p = probs_trg.detach()
        seen["probs"] = p[:, :, ::12, ::12].clone().numpy()
        mx, am = p.max(1)
        srt = torch.sort(p, dim=1)[0]
        seen["argmax"] = am[:, ::4, ::4].to(torch.uint8).numpy()
        seen["margin"] = (srt[:, -1] - srt[:, -2])[:, ::4, ::4].to(torch.float16).numpy()
        seen["confident"] = float((mx >= 0.968).double().mean())
        out = real_mix(images_trg, probs_trg, images_src, gt_src)
        seen["mixed_lbl"] = out[1][:, ::4, ::4].to(torch.uint8).numpy()
        seen["mixed_weight"] = out[2][:, ::8, ::8].clone().numpy()
        return out
    real_refine = model.refine

    def refine(logits_trg, logits_ref, warp_mask, certs):
        # refine() is discontinuous where an INPUT's arg-max is a tie: the static-class mask M follows the arg-max of the target's
        # and of the warped reference's probabilities (segmentation_model.py:446-460); keep both top-2 margins for the test
        mg = []
        for lg in (logits_trg, logits_ref):
            srt = torch.sort(torch.softmax(lg, dim=1), dim=1)[0]
            mg.append(srt[:, -1] - srt[:, -2])
        seen["in_margin"] = torch.minimum(*mg)[:, ::4, ::4].to(torch.float16).numpy()
        return real_refine(logits_trg, logits_ref, warp_mask, certs)
    model._rec.step, model.get_dacs_mix, model.refine = step, mix, refine
    import time
    t0 = time.perf_counter()
    model.training_step(batch, 0)
    dt = time.perf_counter() - t0
    print(f"    reference step at {H}x{W}: {dt:.1f} s on {torch.get_num_threads()} threads")
    ema = float(sum(p.double().abs().sum() for p in model.ema_parameters()))
    live = float(sum(p.double().abs().sum() for p in model.live_parameters()))
    save(name, losses=np.array([model.logged["train_loss_src"], model.logged.get("train_loss_featdist_src", 0.0),
                                model.logged["train_loss_uda_trg"]]), enable_fdist=np.bool_(enable_fdist),
         grad_norms=np.array(model._rec.norms), ema_abs_sum=ema, live_abs_sum=live, size=np.array([H, W]), blk=np.int64(blk),
         cpu_seconds=np.float32(dt), grad_conv_seg=grads["conv_seg"], grad_fc1=grads["fc1"],
         w_q=model.backbone.block1[0].attn.q.weight.detach().flatten()[::61].numpy(),
         w_fuse=model.head.fuse_layer.bottleneck.conv.weight.detach().flatten()[::9973].numpy(),
         probs_sample=seen["probs"], probs_argmax=seen["argmax"], probs_margin=seen["margin"], in_margin=seen["in_margin"],
         confident=np.float64(seen["confident"]), mixed_lbl=seen["mixed_lbl"], mixed_weight=seen["mixed_weight"])
    print("   ", model.logged, model._rec.norms, "confident", seen["confident"])
    torch.set_grad_enabled(False)


def g13_adapt():
    """G13-A: the `adapt_to_ref: True` branch (refign_hrda_star.yaml:92, segmentation_model.py:194-213): with the coin on
    heads the teacher sees the reference image alone -- no align, no refine, pseudo-labels from its plain softmax.  THREE
    consecutive reference steps of the HRDA mit_b0 model (128 x 128, b = 2) from one seed chosen so that both sides of the coin
    occur; per step: which side, the three losses, the group gradient norms; after the last step the EMA / student
    checksums.  The draws of a step: two crop offsets, the coin, the DACS parameters, two crop offsets (python `random`),
    the class choice (torch)."""
    torch.set_grad_enabled(True)
    sm = R.ref_module("models.segmentation_model")
    for seed in range(81, 200):
        random.seed(seed)
        coins = []
        for _ in range(3):                     # a dry run of the python stream: 2 crop draws, coin, jitter, 2 crop draws
            random.randrange(0, 8); random.randrange(0, 8)
            coins.append(random.random() < 0.5)
            random.uniform(0, 1)
            random.randrange(0, 8); random.randrange(0, 8)
        if coins == [True, False, True]:
            break
    model = run_reference(True, adapt_to_ref=True)
    H = W = 128
    batch = make_batch(2, H, W, 64)
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    model.global_step = 3
    aligned = []
    real_align = model.align

    def align(*a, **k):
        aligned[-1] = True
        return real_align(*a, **k)
    model.align = align
    losses, norms = [], []
    for _ in range(3):
        aligned.append(False)
        model.training_step(batch, 0)
        model.global_step += 1
        losses.append([model.logged["train_loss_src"], model.logged["train_loss_featdist_src"], model.logged["train_loss_uda_trg"]])
        norms.append(list(model._rec.norms))
    took_ref = [not a for a in aligned]
    assert took_ref == coins, (took_ref, coins)
    ema = float(sum(p.double().abs().sum() for p in model.ema_parameters()))
    live = float(sum(p.double().abs().sum() for p in model.live_parameters()))
    save("step_hrda_adapt_to_ref", seed=np.int64(seed), adapted_to_ref=np.array(took_ref), losses=np.array(losses),
         grad_norms=np.array(norms), ema_abs_sum=ema, live_abs_sum=live, size=np.array([H, W]))
    print("    seed", seed, "adapted to ref:", took_ref, losses)
    torch.set_grad_enabled(False)


def g13_k4f():
    g13_k4(1088, 1920, True, "step_hrda_b5_1088x1920", 81, 64)


GROUPS = {"G13": g13, "G13B5": g13_b5, "G13K3": g13_k3, "G13K4": g13_k4, "G13K4F": g13_k4f, "G13A": g13_adapt}
