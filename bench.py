#!/usr/bin/env python3
"""bench.py -- Refign align-and-refine hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks (it re-executes itself under
torch.distributed.run on 127.0.0.1, one rank per GPU; it refuses when the node has fewer than N GPUs) -- the form the
reference takes its ranks in (`--trainer.gpus N`, README.md:262).  Under an external torchrun it is one of the ranks.

One "step" = one pass of the hot path over one batch of b=2 synthetic (target, reference) image pairs per GPU at
1080x1920 (BASELINE.json's metric configuration).  Image pairs are independent, so ranks shard pairs with no data-path
collective ("scaling": "weak").  The timed region is bracketed by a barrier + torch.cuda.synchronize() on both sides, the
MAX over ranks is taken, rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant kernel (patch-9 local correlation at level 1: C=128, 270x480 per image), algorithmic bytes
                4*B*H*W*(2C+81) per launch / average launch duration measured live with HIP events on the launch stream.
  cpu_baseline  the reference's CPU path for the same step on a bounded sample (1 pair), timed on this box's host
                cores at N=1 on rank 0: kind "reference" = the reference's own correlation.cpp compiled into oracle/_ref
                (+ torch-CPU ATen ops, which are what the reference's CPU path calls for warp/refine), else kind "port"
                = oracle/ restatement.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (without it RCCL / tensor sharing across processes fails with
# hipIpcGetMemHandle: invalid argument); the image exports it already -- kept here for a launcher that scrubs the environment
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="refign_hrda_step_1080x1920")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "k5"],
                    help="segmentation networks: bf16 autocast (the reference trains with --trainer.precision 16) or "
                         "fp32; the align/refine kernels are always fp32 (the reference forces fp32 there too); k5 = "
                         "BASELINE.json config 5: bf16 + the EMA teacher's MiT blocks on the fp8 (e4m3) matrix-core "
                         "kernels (refign_amd/f8.py)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--pairs-per-gpu", type=int, default=2)
    ap.add_argument("--adapt-to-ref", action="store_true",
                    help="adapt_to_ref: True as refign_hrda_star.yaml:92 writes it: a coin per step, on heads the teacher sees "
                         "the reference image alone (no align, no refine).  Default off = every step aligns and refines (the "
                         "more expensive side of the coin, every step)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-sample", default="full", choices=["full", "small"],
                    help="cpu_baseline leg: one step at the full image size (default, ~1 min) or two reduced sizes + extrapolation")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="ONLY time the CPU baseline at the full image size (minutes) and print it as one JSON line")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo only with --workload launcher_selftest (tests/test_bench_cpu.py)")
    args = ap.parse_args()
    if args.workload == "uawarpc_align_512x512" and (args.height, args.width) == (1080, 1920):
        args.height = args.width = 512                    # K2's own size unless another one is asked for
    return args


# ------------------------------------------------------------------------------------------------------------------
# workload: the HIP-kernel part of align+refine at 1080x1920 (K4 shapes): three local correlation
# layers with on-the-fly warp, the global correlation layer, the fused align tail (upsample + confidence + logits warp)
# and refine.  Synthetic inputs: unit-norm random features, 5 px random flows, N(0,9) logits.
# ------------------------------------------------------------------------------------------------------------------
class AlignRefineKernels:
    name = "align_refine_kernels_1080x1920"

    def __init__(self, dev, b, seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        H, W = 1080, 1920
        self.b, self.H, self.W = b, H, W

        def feat(c, h, w):
            return torch.nn.functional.normalize(torch.randn(b, c, h, w, generator=g), dim=1).to(dev)

        def flow(h, w):
            # flows on the align path are bilinear up-samplings of the coarser level (uawarpc.py:140,213,238) plus a
            # small residual: smooth 5 px field + 0.5 px noise
            coarse = 5.0 * torch.randn(b, 2, max(h // 8, 2), max(w // 8, 2), generator=g)
            f = torch.nn.functional.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=False)
            return (f + 0.5 * torch.randn(b, 2, h, w, generator=g)).to(dev)

        self.c11, self.c21 = feat(128, H // 4, W // 4), feat(128, H // 4, W // 4)
        self.c12, self.c22 = feat(256, H // 8, W // 8), feat(256, H // 8, W // 8)
        self.c13, self.c23 = feat(256, 32, 32), feat(256, 32, 32)
        self.c14, self.c24 = feat(512, 16, 16), feat(512, 16, 16)
        self.f1, self.f2, self.f3 = flow(H // 4, W // 4), flow(H // 8, W // 8), flow(32, 32)
        self.logvar = (2.0 * torch.randn(b, 1, H // 4, W // 4, generator=g)).to(dev)
        self.logits_trg = (3.0 * torch.randn(b, 19, H, W, generator=g)).to(dev)
        self.logits_ref = (3.0 * torch.randn(b, 19, H, W, generator=g)).to(dev)

    def step(self):
        from refign_amd.correlation import local_correlation_layer
        from refign_amd.matching import align_tail
        from refign_amd.modules import GlobalFeatureCorrelationLayer
        from refign_amd.refine import refine
        corr4 = GlobalFeatureCorrelationLayer()(self.c24, self.c14)
        corr3 = local_correlation_layer(self.c23, self.c13, flow=self.f3)
        corr2 = local_correlation_layer(self.c22, self.c12, flow=self.f2)
        corr1 = local_correlation_layer(self.c21, self.c11, flow=self.f1)
        warped, mask, cert = align_tail(self.logits_ref, self.f1 * 4.0, self.logvar)
        probs = refine(self.logits_trg, warped, mask, cert, gamma=0.25)
        return corr4, corr3, corr2, corr1, probs

    # ---- dominant kernel for the roofline: level-1 local correlation (patch 9, fused ReLU + L2 norm) ----
    def roofline_launch(self):
        from refign_amd.correlation import local_correlation_layer
        return local_correlation_layer(self.c21, self.c11)

    def roofline_bytes(self):
        b, c, h, w = self.c11.shape
        return 4 * b * h * w * (2 * c + 81)          # 4*B*H*W*(2C+81)

    # ---- CPU baseline: same step, 1 pair, reference code path on the host ----
    def cpu_step(self, kind, corr_fn):
        import torch.nn.functional as F
        tc = lambda t: t[:1].cpu()  # noqa: E731

        def warp_cpu(x, flo):
            B, C, H, W = x.shape
            xx = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W).expand(B, 1, H, W)
            yy = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1).expand(B, 1, H, W)
            v = torch.cat((xx, yy), 1) + flo
            vx = 2.0 * v[:, 0] / max(W - 1, 1) - 1.0
            vy = 2.0 * v[:, 1] / max(H - 1, 1) - 1.0
            grid = torch.stack((vx, vy), dim=3)
            out = F.grid_sample(x, grid, align_corners=True, padding_mode="zeros")
            return out, (vx > -1) & (vy > -1) & (vx < 1) & (vy < 1)

        def local(src, trg, flo):
            w, _ = warp_cpu(src, flo)
            c = corr_fn(trg.contiguous(), w.contiguous())
            return F.normalize(F.relu(c.reshape(c.shape[0], 81, *c.shape[-2:])), dim=1)

        t0 = time.perf_counter()
        fs, ft = tc(self.c24).flatten(2), tc(self.c14).flatten(2)
        corr = torch.bmm(ft.transpose(1, 2), fs).transpose(1, 2)           # (b, S, T)
        cb = corr / (corr.max(dim=1, keepdim=True)[0] + 1e-5)
        ca = corr / (corr.max(dim=2, keepdim=True)[0] + 1e-5)
        F.normalize(F.relu(corr * (ca * cb)), dim=1)
        local(tc(self.c23), tc(self.c13), tc(self.f3))
        local(tc(self.c22), tc(self.c12), tc(self.f2))
        local(tc(self.c21), tc(self.c11), tc(self.f1))
        fu = F.interpolate(tc(self.f1) * 4.0, size=(self.H, self.W), mode="bilinear", align_corners=False)
        lu = F.interpolate(tc(self.logvar), size=(self.H, self.W), mode="bilinear", align_corners=False)
        cert = 1.0 - torch.exp(-1.0 / (2 * torch.exp(lu)))
        warped, mask = warp_cpu(tc(self.logits_ref), fu)
        lt = tc(self.logits_trg)
        pt, pr = F.softmax(lt, 1), F.softmax(warped, 1)
        ent = -(pt * F.log_softmax(lt, 1)).sum(1) / torch.log(torch.tensor(19.0))
        s = ent.mean(dim=(1, 2)) ** 0.25
        at, ar = pt.argmax(1), pr.argmax(1)
        static = torch.tensor([0, 1, 2, 3, 4, 8, 9, 10])
        M = (torch.isin(at, static) & torch.isin(ar, static)).unsqueeze(1).expand_as(pt).clone()
        M[:, 5:8] = 0
        M[:, 11:] = 0
        eps = s.view(-1, 1, 1, 1) * torch.maximum(cert.expand_as(pt), M.float())
        eps = eps * mask.unsqueeze(1)
        _ = (1 - eps) * pt + eps * pr
        return time.perf_counter() - t0



# ------------------------------------------------------------------------------------------------------------------
# workload: ONE FULL Refign training step (the metric's "align+seg fwd+bwd"): source fwd+bwd, ImageNet feature
# distance, EMA, teacher fwd on (trg, ref), align (VGG-16 + UAWarpC + logits warp), refine, DACS mix, mixed fwd+bwd,
# gradient all-reduce, AdamW + LR schedule -- HRDA MiT-B5 as configs/cityscapes_darkzurich/refign_hrda_star.yaml builds
# it (random init: no checkpoints offline), b=2 source + 2 (target, reference) pairs per GPU at 1080x1920.
# The Refign branch is pinned (adapt_to_ref=False): the reference skips align/refine on a random 50 % of steps.
# ------------------------------------------------------------------------------------------------------------------
REF_CFG = {  # the `model:` / `optimizer:` / `lr_scheduler:` sections of refign_hrda_star.yaml:83-190, verbatim values
    "model": {"class_path": "models.DomainAdaptationSegmentationModel", "init_args": {
        "backbone_lr_factor": 0.1, "enable_fdist": True, "use_hrda": True, "hrda_output_stride": 4,
        "use_slide_inference": True, "use_refign": True, "adapt_to_ref": True, "gamma": 0.25,
        "backbone": {"class_path": "models.backbones.MixVisionTransformer",
                     "init_args": {"model_type": "mit_b5", "pretrained": "cityscapes"}},
        "head": {"class_path": "models.heads.DAFormerHead", "init_args": {
            "in_channels": [64, 128, 320, 512], "in_index": [0, 1, 2, 3], "num_classes": 19,
            "input_transform": "multiple_select"}},
        "hrda_scale_attention": {"class_path": "models.heads.SegFormerHead", "init_args": {
            "in_channels": [64, 128, 320, 512], "in_index": [0, 1, 2, 3], "num_classes": 19,
            "input_transform": "multiple_select"}},
        "alignment_backbone": {"class_path": "models.backbones.VGG", "init_args": {
            "model_type": "vgg16", "pretrained": "imagenet", "out_indices": [2, 3, 4]}},
        "alignment_head": {"class_path": "models.heads.UAWarpCHead", "init_args": {
            "in_index": [0, 1], "input_transform": "multiple_select", "estimate_uncertainty": True,
            "pretrained": "pretrained_models/uawarpc_megadepth.ckpt"}},
        "loss": {"class_path": "models.losses.PixelWeightedCrossEntropyLoss"}}},
    "optimizer": {"class_path": "torch.optim.AdamW", "init_args": {"lr": 0.0006, "weight_decay": 0.01}},
    "lr_scheduler": {"class_path": "helpers.lr_scheduler.LinearWarmupPolynomialLR", "init_args": {
        "warmup_iters": 1500, "warmup_ratio": 0.000001, "power": 1.0, "max_steps": 40000}},
}


PIPELINE_NEXT_BATCH = True


class RefignStep:
    name = "refign_hrda_step_1080x1920"
    use_hrda = True
    # two untimed set-up steps before the W warm-up steps: the first one selects library solvers and fills the weight /
    # constant caches, the second one captures the hipGraphs of the gradient-free half (refign_amd/graphs.py)
    prime_steps = 2

    def __init__(self, dev, b, seed, H=1080, W=1920, precision="bf16", sync_bn=True, adapt_to_ref=False):
        import copy
        import random
        import numpy as np
        from refign_amd import config
        from refign_amd.trainer import Trainer
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        cfg = copy.deepcopy(REF_CFG)
        cfg["model"]["init_args"]["use_hrda"] = self.use_hrda
        if not self.use_hrda:
            cfg["model"]["init_args"].pop("hrda_scale_attention")
        # offline: no checkpoints -> random init of the same architectures; pin the Refign branch
        over = {"backbone.init_args.pretrained": None, "alignment_backbone.init_args.pretrained": None,
                "alignment_head.init_args.pretrained": None, "adapt_to_ref": bool(adapt_to_ref)}
        self.model = config.build_model(cfg, over).to(dev).train()
        self.adapt_to_ref = bool(adapt_to_ref)
        self.trainer = Trainer(self.model, sync_batchnorm=sync_bn)
        self.model.teacher_f8 = precision == "k5"                   # K5: EMA-teacher backbone on the fp8 kernels
        self.precision = precision = "bf16" if precision == "k5" else precision
        self.b, self.H, self.W = b, H, W
        self.name = type(self).name.replace("1080x1920", f"{H}x{W}")     # the label says the size that ran
        g = torch.Generator(device="cpu").manual_seed(seed)
        lbl = torch.randint(0, 19, (b, (H + 31) // 32, (W + 31) // 32), generator=g)
        lbl = lbl.repeat_interleave(32, 1).repeat_interleave(32, 2)[:, :H, :W].contiguous()
        lbl[torch.rand(b, H, W, generator=g) < 0.05] = 255
        trg = torch.randn(b, 3, H, W, generator=g)
        ref = 0.8 * torch.roll(trg, (3, -5), (2, 3)) + 0.2 * torch.randn(b, 3, H, W, generator=g)
        self.batch = {"image_src": torch.randn(b, 3, H, W, generator=g).to(dev), "semantic_src": lbl.to(dev),
                      "image_trg": trg.to(dev), "image_ref": ref.to(dev)}
        self._kern = None

    def step(self):
        # next_batch: what a prefetching loader hands over one step early (here the same synthetic batch) -- the frozen
        # ImageNet encoder's features of the next source images are then computed while this step's mixed pass runs
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.precision == "bf16"):
            self.trainer.step(self.batch, next_batch=self.batch if PIPELINE_NEXT_BATCH else None)

    # roofline kernel + CPU baseline are those of the align/refine kernel workload at the same image size
    def _kernels(self):
        if self._kern is None:
            self._kern = AlignRefineKernels(self.batch["image_src"].device, self.b, 4321)
        return self._kern

    def _level1_features(self):
        """The level-1 operands the step itself feeds the kernel: VGG-16 pool-2 features (128 channels, 1/4 resolution) of
        this workload's (reference, target) images, L2-normalised over channels, as UAWarpCHead.forward does
        (uawarpc.py:96-108) -- the SAME data the in-step launches see, so the HIP-event figure of the bench line and the
        in-step rocprofv3 kernel-trace average (profiles/) measure one thing."""
        if getattr(self, "_l1", None) is None:
            from refign_amd import align as A
            from refign_amd.matching import l2_normalize_channels
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.precision == "bf16"):
                dt = A.align_compute_dtype()
                with torch.autocast("cuda", enabled=dt != torch.float32, dtype=dt if dt != torch.float32 else None):
                    pt, pr, _, _ = A.extract_pyramids(self.model.alignment_backbone, self.batch["image_ref"].float(),
                                                      self.batch["image_trg"].float())
                self._l1 = (l2_normalize_channels(pr[0]), l2_normalize_channels(pt[0]))
        return self._l1

    def roofline_launch(self):
        from refign_amd.correlation import local_correlation_layer
        src, trg = self._level1_features()
        return local_correlation_layer(src, trg)

    def roofline_bytes(self):
        b, c, h, w = self._level1_features()[0].shape
        return 4 * b * h * w * (2 * c + 81)          # 4*B*H*W*(2C+81)

    def cpu_step(self, kind, corr_fn):
        return self._kernels().cpu_step(kind, corr_fn)


class RefignDAFormerStep(RefignStep):
    name = "refign_daformer_step_1080x1920"
    use_hrda = False


class RefignAlignRefine(RefignStep):
    """The gradient-free half of the step only -- what the north star's "image-pairs/s align+refine" counts: EMA-teacher
    forward on (target, reference), align (VGG-16 pyramid, UAWarpC head, logits warp), refine, pseudo-labels
    (segmentation_model.py:194-224).  Secondary workload: `--workload refign_align_refine_1080x1920`."""
    name = "refign_align_refine_1080x1920"

    @torch.no_grad()
    def step(self):
        m, trg, ref = self.model, self.batch["image_trg"], self.batch["image_ref"]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.precision == "bf16"):
            return torch.max(m._teacher_align_refine(trg, ref), dim=1)


class UAWarpCAlign:
    """K2 (BASELINE.json config 2: "UAWarpC align-only, MegaDepth config, 512x512 pairs"): AlignmentModel.forward
    (models/alignment_model.py:55-79) -- VGG-16 pyramids of both images at the input size and at 256 x 256, the UAWarpC head
    (global + three local correlation levels, flow decoders, uncertainty), flow and 1 - P_R up-sampled to the input size -- on
    b pairs per GPU, random-init networks (no checkpoints offline), the reference's AMP recipe (convolutions fp16, correlation /
    warp / uncertainty fp32).  `--workload uawarpc_align_512x512`."""
    name = "uawarpc_align_512x512"
    use_hrda = False

    def __init__(self, dev, b, seed, H=512, W=512, precision="bf16"):
        from refign_amd.align import VGG, UAWarpCHead
        from refign_amd.alignment_model import AlignmentModel
        torch.manual_seed(seed)
        self.model = AlignmentModel(alignment_backbone=VGG('vgg16', out_indices=[2, 3, 4]),
                                    alignment_head=UAWarpCHead(in_index=[0, 1], input_transform='multiple_select',
                                                               estimate_uncertainty=True)).to(dev).eval()
        self.precision = "bf16" if precision == "k5" else precision
        self.b, self.H, self.W = b, H, W
        self.name = f"uawarpc_align_{H}x{W}"
        g = torch.Generator(device="cpu").manual_seed(seed)
        img_i = torch.randn(b, 3, H, W, generator=g)
        self.img_i = img_i.to(dev)
        self.img_j = (0.8 * torch.roll(img_i, (3, -5), (2, 3)) + 0.2 * torch.randn(b, 3, H, W, generator=g)).to(dev)
        self._l1 = None

    @torch.no_grad()
    def step(self):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.precision == "bf16"):
            return self.model(self.img_i, self.img_j)

    def _level1_features(self):
        if self._l1 is None:
            from refign_amd import align as A
            from refign_amd.matching import l2_normalize_channels
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.precision == "bf16"):
                dt = A.align_compute_dtype()
                with torch.autocast("cuda", enabled=dt != torch.float32, dtype=dt if dt != torch.float32 else None):
                    pt, pr, _, _ = A.extract_pyramids(self.model.alignment_backbone, self.img_j.float(), self.img_i.float())
                self._l1 = (l2_normalize_channels(pr[0]), l2_normalize_channels(pt[0]))
        return self._l1

    def roofline_launch(self):
        from refign_amd.correlation import local_correlation_layer
        return local_correlation_layer(*self._level1_features())

    def roofline_bytes(self):
        b, c, h, w = self._level1_features()[0].shape
        return 4 * b * h * w * (2 * c + 81)

    def cpu_pairs_per_s(self, cores):
        """The same forward on the host: oracle/cpu_align.alignment_forward (pinned against the reference's output,
        tests/test_oracle_cpu.py) on this workload's b pairs, fp32."""
        import copy
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import cpu_align
        kind, corr_fn = cpu_align._corr_fn_default()
        m = copy.deepcopy(self.model).cpu().float()
        i, j = self.img_i.cpu(), self.img_j.cpu()
        cpu_align.alignment_forward(m.alignment_backbone, m.alignment_head, i[:1], j[:1], corr_fn)      # thread pool / allocator
        t0 = time.perf_counter()
        cpu_align.alignment_forward(m.alignment_backbone, m.alignment_head, i, j, corr_fn)
        dt = time.perf_counter() - t0
        return {"value": round(self.b / dt, 4), "unit": "image-pairs/s", "cores": cores,
                "kind": "restatement+reference-correlation" if kind == "reference" else "port",
                "sample": f"AlignmentModel.forward for {self.b} pairs at {self.H}x{self.W} on the host with {cores} threads, fp32: "
                          f"{dt:.2f} s wall (oracle/cpu_align.alignment_forward; correlation = "
                          f"{'reference correlation.cpp (oracle/_ref)' if kind == 'reference' else 'oracle/corr_oracle.c'})"}


class LauncherSelfTest:
    """No kernels: a CPU stand-in for a workload so that the launcher half of this file (self-launch of N ranks, rendezvous on
    127.0.0.1, barrier + max-over-ranks timing, the one JSON line of rank 0) can be exercised without a GPU
    (tests/test_bench_cpu.py, backend gloo).  Never a measurement."""
    name = "launcher_selftest"
    use_hrda = False

    def __init__(self, *a, **k):
        self.x = torch.ones(1024)

    def step(self):
        self.x = self.x * 1.0001


def pmc_traffic():
    """HBM-side bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per
    MI355X_MICROARCH.md; separate passes) -- the latest profiles/rNN_pmc_traffic_corr9.json; null if absent."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_corr9.json")), reverse=True):
        try:
            with open(path) as f:
                return int(json.load(f)["hbm_traffic_bytes_per_launch"])
        except Exception:
            continue
    return None


def rocprof_in_step_us():
    """Average duration of the roofline kernel INSIDE the timed steps from the committed rocprofv3 kernel trace of this
    command (profiles/rNN_rocprofv3_bench_kernel_stats_timed_region.csv, written by tools/final_check.sh): there the
    launch sits between other kernels, the HIP-event figure of this run is 20 launches back to back (a sustained fp32
    load at a lower clock: DESIGN.md section 4.2).  Reported next to it, never instead of it; null if absent."""
    import csv
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_bench_kernel_stats_timed_region.csv")),
                       reverse=True):
        try:
            with open(path) as f:
                for r in csv.DictReader(f):
                    # round 4: the software-pipelined 4-stage-ring kernel is the level-1 default; older summaries hold its
                    # predecessors
                    if any(k in r["Name"] for k in ("corr9_pipe2_kernel<16, 32", "corr9_pipe_kernel<16, 32",
                                                    "corr9_dma_kernel<16, 32")):
                        return {"avg_launch_us": round(float(r["AverageNs"]) / 1e3, 2), "launches": int(r["Calls"]),
                                "source": os.path.relpath(path, ROOT)}
        except Exception:
            continue
    return None


WORKLOADS = {AlignRefineKernels.name: AlignRefineKernels, RefignStep.name: RefignStep,
             RefignDAFormerStep.name: RefignDAFormerStep, RefignAlignRefine.name: RefignAlignRefine,
             UAWarpCAlign.name: UAWarpCAlign, LauncherSelfTest.name: LauncherSelfTest}


def cpu_baseline(wl, args):
    """The reference's CPU path for the SAME step on the host cores, on a bounded sample: ONE pair at a reduced image
    size (the step's cost is linear in pixels), extrapolated to the metric's unit.  kind "reference" when the
    reference's own correlation.cpp (oracle/_ref) is available, else "port"; everything else is the reference's
    algorithm restated on torch-CPU ops (oracle/cpu_align.py + the same module trees on CPU)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    # threads: all host cores up to 32 -- the step is thousands of small ops, and with 256 OpenMP threads per op the
    # fork/join overhead dominates (measured: >7 min for one step on the 256-core box vs tens of seconds at 32)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    import cpu_align
    kind, corr_fn = cpu_align._corr_fn_default()
    if isinstance(wl, UAWarpCAlign):
        return wl.cpu_pairs_per_s(cores)
    if isinstance(wl, AlignRefineKernels):
        dt = wl.cpu_step(kind, corr_fn)
        return {"value": round(1.0 / dt, 5), "unit": "image-pairs/s", "cores": cores, "kind": kind,
                "sample": f"1 pair through the align+refine kernel stage on the host, {dt:.2f} s wall"}
    import copy
    import math
    from refign_amd import config
    from refign_amd.trainer import Trainer
    cfg = copy.deepcopy(REF_CFG)
    cfg["model"]["init_args"]["use_hrda"] = wl.use_hrda
    if not wl.use_hrda:
        cfg["model"]["init_args"].pop("hrda_scale_attention")
    over = {"backbone.init_args.pretrained": None, "alignment_backbone.init_args.pretrained": None,
            "alignment_head.init_args.pretrained": None, "adapt_to_ref": False}
    model = config.build_model(cfg, over).train()
    model.align = lambda lr, ir, it: cpu_align.align(model.alignment_backbone, model.alignment_head, lr, ir, it, corr_fn)
    model.refine = lambda lt, lr, m, c: cpu_align.refine(lt, lr, m, c, gamma=model.gamma)
    trainer = Trainer(model, fused_optimizer=False)

    def one_step(h, w):
        g = torch.Generator().manual_seed(99)
        lbl = torch.randint(0, 19, (1, (h + 31) // 32, (w + 31) // 32), generator=g)
        lbl = lbl.repeat_interleave(32, 1).repeat_interleave(32, 2)[:, :h, :w].contiguous()
        trg = torch.randn(1, 3, h, w, generator=g)
        batch = {"image_src": torch.randn(1, 3, h, w, generator=g), "semantic_src": lbl, "image_trg": trg,
                 "image_ref": 0.8 * torch.roll(trg, (1, -1), (2, 3)) + 0.2 * torch.randn(1, 3, h, w, generator=g)}
        t0 = time.perf_counter()
        trainer.step(batch)
        return time.perf_counter() - t0

    full_px = float(args.height * args.width)
    detail = ("restatement + reference correlation: the reference's compiled correlation.cpp, every other op the reference's "
              "algorithm restated on torch-CPU") if kind == "reference" else "restatement (oracle/corr_oracle.c + torch-CPU ops)"
    kind_out = "restatement+reference-correlation" if kind == "reference" else "port"
    if getattr(args, "cpu_baseline_full", False) or getattr(args, "cpu_sample", "full") == "full":
        # the metric's own unit of work, measured: one pair at the full image size (55 s on the 32 host threads of the MI355X
        # box in round 4 -- VERDICT r3 asked for the measurement instead of an extrapolation from 6 % of the pixels)
        dt = one_step(args.height, args.width)
        return {"value": round(1.0 / dt, 6), "unit": "image-pairs/s", "cores": cores, "kind": kind_out, "kind_detail": detail,
                "sample": f"ONE full training step (same model/config, fp32) for 1 source image + 1 pair at the FULL "
                          f"{args.height}x{args.width} on the host with {cores} threads: {dt:.1f} s wall.  Correlation = "
                          f"{'reference correlation.cpp (oracle/_ref)' if kind == 'reference' else 'oracle/corr_oracle.c'}"
                          f" + OpenMP, all other ops torch-CPU ATen (what the reference's CPU path calls)"}
    # --cpu-sample small: two reduced sizes (multiples of 32: HRDA crop boxes need H/2 and W/2 divisible by 16), the exponent
    # of time against pixels fitted from them, the larger one extrapolated to the full size with that exponent.  (Round 4:
    # exponent 0.78 between 6 % and 13 % of the pixels -- fixed per-op costs still weigh at those sizes -- and the
    # extrapolation is then 47 % too fast against the measured full size: hence the full size is the default.)
    sizes = [(max(64, args.height // 4 // 32 * 32), max(64, args.width // 4 // 32 * 32)),
             (max(96, args.height * 3 // 8 // 32 * 32), max(96, args.width * 3 // 8 // 32 * 32))]
    one_step(64, 64)                                     # warm the allocator / thread pool: not part of either sample
    times = [one_step(h, w) for h, w in sizes]
    px = [float(h * w) for h, w in sizes]
    expo = math.log(times[1] / times[0]) / math.log(px[1] / px[0]) if px[1] > px[0] and times[0] > 0 else 1.0
    t_full = times[1] * (full_px / px[1]) ** expo
    return {"value": round(1.0 / t_full, 6), "unit": "image-pairs/s", "cores": cores, "kind": kind_out, "kind_detail": detail,
            "fitted_exponent": round(expo, 3),
            "samples": [{"size": f"{h}x{w}", "pixel_fraction": round(p_ / full_px, 4), "seconds": round(t, 2)}
                        for (h, w), p_, t in zip(sizes, px, times)],
            "sample": f"EXTRAPOLATED: one full training step for 1 source image + 1 pair on the host with {cores} threads at "
                      f"{sizes[0][0]}x{sizes[0][1]} ({times[0]:.2f} s) and {sizes[1][0]}x{sizes[1][1]} ({times[1]:.2f} s): time ~ "
                      f"pixels^{expo:.2f}; value = 1 pair / ({times[1]:.2f} s x ({full_px / px[1]:.2f})^{expo:.2f})"}


def launch_series_us(launch, spaced, reps=20):
    """Average duration of `launch()` from HIP events around EACH launch on the current stream; `spaced`: every launch behind
    ~1 ms of an idle device (one spinning wave) -- see the roofline block of main().  Shared with tools/kbench.py so that the
    level-1 correlation rows there and the bench line are ONE measurement (same operands, same timing)."""
    pairs = []
    torch.cuda.synchronize()
    for _ in range(reps):
        if spaced:
            torch.cuda._sleep(2_000_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a_.elapsed_time(b_) for a_, b_ in pairs) * 1e3 / reps


def _stall_guard(rank, world, progress):
    """N > 1 only: if a multi-rank run makes no progress for RFN_STALL_S (RFN_BENCH_STALL_S) seconds (default 600), say where it
    stopped and exit non-zero instead of hanging the node.  The guard itself lives in the trainer (refign_amd.trainer.StallGuard,
    which also watches Trainer.step in any other driver); `progress` is its [time, phase] heartbeat, updated by tick()."""
    from refign_amd.trainer import StallGuard
    g = StallGuard(rank, world)
    g.progress = progress
    return g


def _self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: become N ranks.  One node, one rank per GPU, rendezvous on 127.0.0.1
    (the container's hostname may not resolve) at a free port; the children see RANK / LOCAL_RANK / WORLD_SIZE and take the
    ordinary path through main().  Fails loudly when the node cannot give every rank its own GPU."""
    import socket
    import subprocess
    if args.backend == "nccl":
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node has {have} GPU(s); one rank per GPU, no oversubscription")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.cpu_baseline_full:                  # host cores only: one full-size CPU step, one JSON line, done
        import types
        wl = types.SimpleNamespace(use_hrda=WORKLOADS[args.workload].use_hrda)
        print(json.dumps(cpu_baseline(wl, args)), flush=True)
        return
    selftest = args.workload == LauncherSelfTest.name
    if args.backend == "gloo" and not selftest:
        raise SystemExit("--backend gloo is for --workload launcher_selftest only (the product path has no CPU fallback)")
    if not selftest and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    dev = torch.device("cpu")
    if not selftest:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist = None
    progress = [time.monotonic(), "start"]
    if world > 1 or "RANK" in os.environ:     # under torchrun always a process group, also for a single rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            _stall_guard(rank, world, progress)
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if selftest:
        wl = LauncherSelfTest()
    else:
        import refign_amd
        refign_amd.load_library()
        if args.workload == AlignRefineKernels.name:
            wl = AlignRefineKernels(dev, args.pairs_per_gpu, seed=1234 + rank)
        else:
            kw = {"adapt_to_ref": True} if args.adapt_to_ref else {}
            wl = WORKLOADS[args.workload](dev, args.pairs_per_gpu, 1234 + rank, args.height, args.width, args.precision, **kw)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        sync()

    def tick(what):
        progress[0], progress[1] = time.monotonic(), what

    for i in range(getattr(wl, "prime_steps", 0)):     # set-up, not warm-up: solver selection, caches, hipGraph capture
        wl.step()
        if world > 1:                                  # (N > 1 only: keeps the guard's clock honest, off the N=1 path)
            sync()
        tick(f"prime step {i}")
    if getattr(wl, "adapt_to_ref", False):
        # both sides of the coin have to be past their eager warm-up calls and captured before the timed region (untimed
        # set-up like the prime steps; the coin is the step's own draw from python's `random`, seeded per rank above)
        for i in range(24):
            tb = wl.model._graphs["teacher_backbone"].states.values()
            if len(tb) == 2 and all(s_["graph"] is not None or s_["failed"] for s_ in tb) and \
                    all(s_["graph"] is not None or s_["failed"] for s_ in wl.model._graphs["tail_refine"].states.values()):
                break
            wl.step()
            tick(f"adapt_to_ref prime step {i}")
    for i in range(args.warmup):
        wl.step()
        if world > 1:
            sync()
        tick(f"warm-up step {i}")
    barrier()
    tick("timed region")
    a0 = getattr(getattr(wl, "model", None), "_adapted_to_ref_steps", 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    barrier()
    dt = time.perf_counter() - t0
    adapted_in_timed = getattr(getattr(wl, "model", None), "_adapted_to_ref_steps", 0) - a0
    tick("after the timed region")
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    small_ar_us = None
    if dist is not None and not selftest and dev.type == "cuda" and args.backend == "nccl":
        # EVERY rank: what one small exchange costs from the step's stream (reported in config.data_parallel)
        probe_t = torch.zeros(512, device=dev)
        for _ in range(5):
            dist.all_reduce(probe_t)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            dist.all_reduce(probe_t)
        e1.record()
        e1.synchronize()
        small_ar_us = round(e0.elapsed_time(e1) * 1e3 / 50, 1)

    roof = None
    if not args.no_roofline and rank == 0 and not selftest:
        # HIP events around EACH launch of the dominant kernel, on the stream it is launched on.  Two series of 20:
        # "spaced" -- every launch behind ~1 ms of an idle device (one spinning wave), which also hides the host's launch
        # latency: the kernel as it runs in the step, between other work (the rocprofv3 kernel trace of the timed steps
        # reports the same duration, `rocprofv3_in_step`); "back_to_back" -- 20 launches in a row, a sustained fp32 load
        # under which the shader clock drops from ~1.9 to ~1.7 GHz.  `achieved` / `frac` are
        # the spaced series; both are reported.
        reps = 20
        for _ in range(3):
            wl.roofline_launch()

        series = lambda spaced: launch_series_us(wl.roofline_launch, spaced, reps)  # noqa: E731

        us_b2b = series(False)
        us = series(True)
        ach = wl.roofline_bytes() / (us * 1e-6) / 1e9
        rb, rc, rh, rw = (wl._level1_features()[0].shape if hasattr(wl, "_level1_features") else wl.c11.shape)
        roof = {"kernel": "corr9_pipe2_kernel<16x32 tiles, 4-stage LDS-DMA ring, software-pipelined rows, fused ReLU+L2norm> level 1 "
                          f"(C={rc}, {rh}x{rw}, b={rb})",
                "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4),
                # (the committed counter passes and kernel trace are of the 1080x1920 level-1 shape only)
                "traffic": pmc_traffic() if (rh, rw, rb) == (270, 480, 2) else None, "avg_launch_us": round(us, 2),
                "avg_launch_us_back_to_back": round(us_b2b, 2),
                "frac_back_to_back": round(wl.roofline_bytes() / (us_b2b * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                # the same launch against the fp32 vector peak: the kernel is a dot per pixel pair (2 x 81 x C flop per pixel, no matrix
                # pipe) and its packed FMAs, not its bytes, are what it issues most of the launch
                "second_bound": {"bound": "fp32 valu", "achieved": round(2.0 * 81 * rc * rb * rh * rw / (us * 1e-6) / 1e12, 1),
                                 "peak": 157.3, "unit": "TFLOP/s",
                                 "frac": round(2.0 * 81 * rc * rb * rh * rw / (us * 1e-6) / 1e12 / 157.3, 4)},
                "algorithmic_bytes_per_launch": wl.roofline_bytes(), "rocprofv3_in_step": rocprof_in_step_us() if (rh, rw, rb) == (270, 480, 2) else None}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and not selftest:
        cpu = cpu_baseline(wl, args)

    ranks_in_group = world
    if dist is not None:
        ranks_in_group = dist.get_world_size()
        dist.barrier()
        sync()
        if not selftest:
            from refign_amd import rccl
            rccl.destroy_all()                     # our own communicators (SyncBatchNorm exchanges), creation order
        dist.destroy_process_group()

    if rank == 0 and selftest:
        print(json.dumps({"metric": "launcher self-test (no kernels, not a measurement)", "value": 0.0, "unit": "none",
                          "n_gpus": world, "ranks_in_group": ranks_in_group, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3), "config": {"workload": wl.name}}))
        return
    if rank == 0:
        pairs = args.pairs_per_gpu * world * args.steps
        size = f"{getattr(wl, 'H', args.height)}x{getattr(wl, 'W', args.width)}"
        step_kind = isinstance(wl, RefignStep) and type(wl) is not RefignAlignRefine
        line = {
            "metric": {AlignRefineKernels.name: f"image-pairs/s (align+refine HIP kernels only, {size})",
                       RefignAlignRefine.name: f"image-pairs/s (teacher fwd + align + refine, no student fwd/bwd, {size})",
                       UAWarpCAlign.name: f"image-pairs/s (UAWarpC align-only: AlignmentModel.forward, {size})"}.get(
                           args.workload, f"image-pairs/s (align+seg fwd+bwd, {size})"),
            "value": round(pairs / dt, 3), "unit": "image-pairs/s", "n_gpus": world, "ranks_in_group": ranks_in_group,
            "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (3 x bf16 split products)" if args.precision == "fp32" and args.workload != AlignRefineKernels.name
                      else "f32")
                     if (args.workload == AlignRefineKernels.name or args.precision == "fp32") else
                     ("bf16+fp8(e4m3) teacher" if args.precision == "k5" else
                      ("f16 VGG-16, split-bf16 (fp32-class) head convolutions, f32 correlation / warp / uncertainty"
                       if args.workload == UAWarpCAlign.name else "bf16")),
            "data": "synthetic",
            "config": {"workload": wl.name, "pairs_per_gpu": args.pairs_per_gpu, "image": size,
                       "networks": ("VGG-16 + UAWarpC head (random init)" if args.workload == UAWarpCAlign.name else
                                    ("HRDA MiT-B5" if getattr(wl, "use_hrda", True) else "MiT-B5") +
                                    " + DAFormer head + VGG-16/UAWarpC align (random init)"),
                       "precision_map": (("fp32 storage everywhere; Linear / convolution / attention products as three bf16 "
                                          "products on the MFMA kernels (refign_amd/split32.py, ~2^-16 relative)"
                                          )
                                         if args.precision == "fp32" else
                                         ("K5: as the bf16 map, plus the attention half of the EMA teacher's MiT blocks (q / kv / "
                                          "spatial-reduction / proj Linear layers and the attention core, 40 views) on fp8 e4m3 MFMA "
                                          "kernels, the Mix-FFN half on the bf16 kernels (f8.HYBRID_FFN); " if args.precision == "k5"
                                          else "") +
                                         "reference AMP recipe: seg nets bf16 autocast (fp32 master weights, grads, "
                                         "norm statistics, losses); matcher: VGG-16 fp16 autocast, UAWarpC head "
                                         "convolutions as three bf16 split products on fp32 storage (align.HEAD_SPLIT: the "
                                         "1e-3 parity bound holds inside the timed map; plain fp16 there: 5.6e-3), "
                                         "correlation, warp, L2-norm, uncertainty and refine kernels fp32"),
                       "parallelism": (f"dp{world}: pairs sharded, align/refine/teacher replica-local, one flat "
                                       f"gradient all-reduce per step over RCCL") if step_kind else
                                      f"dp{world}: pairs sharded, gradient-free, no collective in the timed region"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if step_kind:
            line["config"]["adapt_to_ref"] = (
                f"True (refign_hrda_star.yaml:92): {adapted_in_timed} of the {args.steps} timed steps fell on heads (teacher on the "
                f"reference image alone, no align / refine), {args.steps - adapted_in_timed} aligned and refined"
                if getattr(wl, "adapt_to_ref", False) else
                "False: every step aligns and refines (the YAML's coin would skip that on half of the steps)")
        if step_kind and (world > 1 or "RANK" in os.environ):
            from refign_amd import bn as _bn
            tr = getattr(wl, "trainer", None)
            gb = getattr(tr, "grads", None)
            line["config"]["data_parallel"] = {
                "mode": getattr(tr, "ddp_mode", None),
                "statistics_exchanges": ("none (one rank: batch statistics are local)" if world == 1 and not _bn.data_parallel() else
                                         "RCCL calls of our own on the pass's stream" if _bn._DIRECT["default"] is not None
                                         else "torch.distributed"),
                "gradient_reduce": ("none (one rank: the all-reduce is the identity)" if world == 1 and not _bn.data_parallel() else
                                    "own communicator + stream, released ranges inside the last backward pass"
                                    if gb is not None and gb._comm is not None else "torch.distributed buckets"),
                "reduced_inside_last_backward": (round(getattr(gb, "overlapped_elements", 0) / gb.flat.numel(), 3)
                                                 if gb is not None else None)}
            if small_ar_us is not None:
                # device time of one tiny all-reduce issued from the step's stream (50 in a row, after the timed region): torch.distributed
                # runs a collective on a stream of its own, and when that stream does not share the issuing stream's hardware queue
                # every exchange is two cross-queue hand-overs (profiles/r06_stream_priority_ab.txt)
                line["config"]["data_parallel"]["small_all_reduce_us"] = small_ar_us
        from refign_amd import mfma as _mfma
        # dense ops that ended up in a ROCm library (hipBLASLt / MIOpen / fused SDPA) instead of a hand-written kernel,
        # by call site and dtype, over the whole run (refign_amd/mfma.py: note_library)
        line["config"]["library_fallbacks"] = _mfma.library_summary()
        graphs = getattr(getattr(wl, "model", None), "_graphs", None)
        if graphs:
            # which regions of the step replay from hipGraphs (a failed capture falls back to eager launches: slower,
            # same results -- visible here instead of only as a warning on stderr)
            st = {k: [("replay" if s_["graph"] is not None else ("eager (capture failed)" if s_["failed"] else "eager"))
                      for s_ in g.states.values()] for k, g in graphs.items()}
            line["config"]["hipgraph_regions"] = {k: (v[0] if len(v) == 1 else v) for k, v in st.items() if v}
            steps_m = getattr(wl.model, "_mixed_concurrent_steps", None)
            if steps_m is not None:
                early = getattr(wl.model, "_mixed_early_forwards", 0)
                line["config"]["mixed_pass"] = (("own stream; forward queued before the pseudo-labels exist, loss + backward behind "
                                                 "the teacher branch" if early else "own stream next to the tail of the source pass")
                                                if steps_m else "in stream order after the source pass")
                probe = getattr(wl.model, "_mix_stream_probe", None)
                if probe is not None:
                    # slow-down of a spin kernel on the candidate stream next to one on the main / side stream, per candidate tried
                    # (~1 = own hardware queue, ~2 = shares one): the last one is the stream in use (graphs.concurrent_stream)
                    line["config"]["mixed_pass_stream_probe"] = probe
        if isinstance(wl, RefignStep) and type(wl) is not RefignAlignRefine:
            line["config"]["next_batch_prefetch"] = (
                "frozen ImageNet-encoder features of the next step's source images and the frozen matcher's flow of the next "
                "(reference, target) pair computed during this step's mixed pass (same work per step, same numbers: "
                "tests/test_step_gpu.py::test_prefetched_imnet_features_give_the_same_trajectory)") if PIPELINE_NEXT_BATCH \
                else "off"
        print(json.dumps(line))


if __name__ == "__main__":
    main()
