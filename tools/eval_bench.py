#!/usr/bin/env python3
"""tools/eval_bench.py -- evaluation forward of the bench model (HRDA MiT-B5, the
refign_hrda_star config: sliding-window inference, 1080x1080 crops, stride 420) on 1080x1920 images: images/s."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--precision", default="bf16")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    wl = bench.RefignStep(dev, 1, 1234)
    model = wl.model.eval()
    print("slide inference:", model.use_slide_inference, model.inference_crop_size, model.inference_stride,
          "batched:", model.inference_batched_slide)
    x = torch.randn(args.b, 3, 1080, 1920, device=dev)
    y = torch.randint(0, 19, (args.b, 1080, 1920), device=dev)

    def step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.precision == "bf16"):
            return model.validation_step({"image": x, "semantic": y}, 0, 0, src_name="")

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host = (time.perf_counter() - t0) / args.steps        # what the host needs to enqueue a batch (== dt: host-bound)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"evaluation forward b={args.b} 1080x1920 ({args.precision}): {dt * 1e3:.1f} ms/batch (host enqueue {host * 1e3:.1f} ms), {args.b / dt:.2f} images/s, "
          f"max mem {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")


if __name__ == "__main__":
    main()
