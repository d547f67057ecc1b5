#!/usr/bin/env python3
"""tools/matcher_bench.py -- matcher TRAINING step at the shapes of
configs/megadepth/uawarpc_stage2.yaml: batch 6, 520x520 crops, VGG-16 + UAWarpCHead, Huber multi-scale flow loss +
W-bipath loss with visibility mask, Adam(lr 5e-5, wd 4e-4).  One step = AlignmentModel.training_step + backward +
optimizer step on synthetic images / flows.  Prints ms/step and image-triplets/s.
    python tools/matcher_bench.py [--steps 10] [--b 6] [--size 520]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refign_amd import config  # noqa: E402

MODEL = {"class_path": "models.AlignmentModel", "init_args": {
    "pretrained": None,
    "alignment_backbone": {"class_path": "models.backbones.VGG",
                           "init_args": {"model_type": "vgg16", "pretrained": None, "out_indices": [2, 3, 4]}},
    "alignment_head": {"class_path": "models.heads.UAWarpCHead",
                       "init_args": {"in_index": [0, 1], "input_transform": "multiple_select",
                                     "estimate_uncertainty": True, "iterative_refinement": True}},
    "selfsupervised_loss": {"class_path": "models.losses.MultiScaleFlowLoss", "init_args": {"loss_type": "HuberLoss"}},
    "unsupervised_loss": {"class_path": "models.losses.WBipathLoss",
                          "init_args": {"objective": "multi_scale_flow_loss", "loss_type": "HuberLoss",
                                        "visibility_mask": True}},
}}
OPTIM = {"class_path": "torch.optim.Adam", "init_args": {"lr": 5e-5, "weight_decay": 4e-4}}
SCHED = {"class_path": "torch.optim.lr_scheduler.MultiStepLR",
         "init_args": {"milestones": [100000, 150000, 200000], "gamma": 0.5}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--b", type=int, default=6)
    ap.add_argument("--size", type=int, default=520)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp16"],
                    help="fp16 = the reference's recipe for matcher training (README.md:289-294: --trainer.precision 16): "
                         "fp16 autocast + loss scaling, correlation / warp / losses in fp32")
    ap.add_argument("--census", action="store_true", help="torch.profiler on one step: ATen operators with device time, by input shape")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = config.build_model({"model": MODEL, "optimizer": OPTIM, "lr_scheduler": SCHED}).to(dev).train()
    (opt,), (sch,) = model.configure_optimizers()
    b, S = args.b, args.size
    g = torch.Generator().manual_seed(1)
    trg = torch.randn(b, 3, S, S, generator=g)
    ref = 0.8 * torch.roll(trg, (3, -5), (2, 3)) + 0.2 * torch.randn(b, 3, S, S, generator=g)
    yy, xx = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing="ij")
    flow = torch.stack((4 + 0.02 * (xx - S / 2) + 3 * torch.sin(2 * torch.pi * yy / S),
                        -3 - 0.015 * (yy - S / 2) + 2 * torch.cos(2 * torch.pi * xx / S))).expand(b, 2, S, S).contiguous()
    batch = {"image_ref": ref.to(dev), "image_trg": trg.to(dev), "flow_prime": flow.to(dev),
             "mask_prime": torch.ones(b, S, S, dtype=torch.bool, device=dev), "prime_trg_idx": [i % 2 for i in range(b)]}
    from refign_amd.matching import warp
    with torch.no_grad():
        srcs = torch.stack([(batch["image_ref"], batch["image_trg"])[k][i] for i, k in enumerate(batch["prime_trg_idx"])])
        batch["image_prime"] = warp(srcs, batch["flow_prime"])

    amp = args.precision == "fp16"
    scaler = torch.amp.GradScaler("cuda", enabled=amp)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            loss = model.training_step(batch, 0)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        sch["scheduler"].step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.census:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            step()
            torch.cuda.synchronize()
        rows = []
        for e in prof.key_averages(group_by_input_shape=True):
            dt = getattr(e, "self_device_time_total", None)
            if dt is None:
                dt = e.self_cuda_time_total
            if dt > 0 and e.key.startswith("aten::"):
                rows.append((dt / 1e3, e.count, e.key, str(e.input_shapes)[:110]))
        rows.sort(reverse=True)
        print(f"# ATen operators with device time in one matcher training step: {sum(r[0] for r in rows):.1f} ms in {sum(r[1] for r in rows)} calls")
        for r in rows[:40]:
            print(f"{r[0]:8.2f} {r[1]:6d}  {r[2]:34s} {r[3]}")
        return
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host = (time.perf_counter() - t0) / args.steps        # what the host needs to enqueue a step (== dt: host-bound)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"matcher training step b={b} {S}x{S} {args.precision}: {dt * 1e3:.1f} ms/step (host enqueue {host * 1e3:.1f} ms), {b / dt:.2f} image-triplets/s, "
          f"loss {float(loss):.3f}, max mem {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")
    from refign_amd import mfma as _mfma
    print("library_fallbacks:", _mfma.library_summary())


if __name__ == "__main__":
    main()
