#!/usr/bin/env python3
"""tools/kbench.py -- per-kernel micro-benchmark (HIP events on the launch stream) for the kernels of the hot path.
Prints one line per kernel: average launch time, algorithmic bytes, GB/s and fraction of the 8 TB/s HBM spec peak."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refign_amd import correlation, matching, refine as refine_mod  # noqa: E402
from refign_amd.modules import GlobalFeatureCorrelationLayer  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=2)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    b = args.b
    g = torch.Generator().manual_seed(0)
    rows = []

    def add(name, us, nbytes, flops=0.0):
        rows.append((name, us, nbytes, flops))
        print(f"{name:78s} {us:10.1f} us  {nbytes / 1e6:9.1f} MB  {nbytes / us / 1e3:8.1f} GB/s  "
              f"{nbytes / us / 1e3 / 8000:6.3f} of 8TB/s  {flops / us / 1e6:7.2f} TFLOP/s", flush=True)

    for (lvl, C, H, W) in [("L1", 128, 270, 480), ("L2", 256, 135, 240), ("L3", 256, 32, 32), ("K2-L1", 128, 128, 128)]:
        if args.only and lvl not in args.only:
            continue
        f1 = torch.nn.functional.normalize(torch.randn(b, C, H, W, generator=g), dim=1).to(dev)
        f2 = torch.nn.functional.normalize(torch.randn(b, C, H, W, generator=g), dim=1).to(dev)
        note = ""
        if lvl == "L1":
            # the bench line's own measurement first: the operands the step feeds the kernel (VGG-16 pool-2 features of the bench
            # workload's image pair, L2-normalised) and bench.py's timing (HIP events per launch, spaced / back to back).  The
            # white-noise rows below are a different operand distribution (denser products, lower clock): labelled as such.
            import bench
            wl = bench.WORKLOADS["refign_hrda_step_1080x1920"](dev, b, 1234, 1080, 1920, "bf16")
            s1, t1 = wl._level1_features()
            nb1 = wl.roofline_bytes()
            for _ in range(3):
                wl.roofline_launch()
            add(f"corr9 +relu+l2norm   L1 step operands, spaced (= bench.py roofline)", bench.launch_series_us(wl.roofline_launch, True), nb1)
            add(f"corr9 +relu+l2norm   L1 step operands, back to back", bench.launch_series_us(wl.roofline_launch, False), nb1)
            add(f"corr9 +relu+l2norm   L1 white noise, spaced", bench.launch_series_us(lambda: correlation.local_correlation_layer(f2, f1), True), nb1)
            del wl, s1, t1
            note = " [white noise, back to back]"
        fl = (5 * torch.randn(b, 2, H, W, generator=g)).to(dev)
        nb = 4 * b * H * W * (2 * C + 81)
        fp = 2.0 * 81 * C * b * H * W
        add(f"corr9 raw            {lvl} C={C} {H}x{W}{note}", timeit(lambda: correlation.forward(f1, f2, 1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)), nb, fp)
        add(f"corr9 +relu+l2norm   {lvl} C={C} {H}x{W}{note}", timeit(lambda: correlation.local_correlation_layer(f2, f1)), nb, fp)
        add(f"corr9 +warp+relu+l2n {lvl} C={C} {H}x{W}{note}", timeit(lambda: correlation.local_correlation_layer(f2, f1, flow=fl)), nb + 8 * b * H * W, fp)
        add(f"warp features        {lvl} C={C} {H}x{W} [white-noise flow, sigma 5 px]", timeit(lambda: matching.warp_nocheck(f2, fl)), 4 * b * H * W * (2 * C + 2))
        # a SMOOTH flow (what a matcher emits: a 1/16-resolution field up-sampled): neighbouring lanes gather neighbouring taps
        fs = torch.nn.functional.interpolate(5 * torch.randn(b, 2, max(H // 16, 2), max(W // 16, 2), generator=g), size=(H, W),
                                             mode="bilinear", align_corners=False).contiguous().to(dev)
        add(f"warp features        {lvl} C={C} {H}x{W} [smooth flow]", timeit(lambda: matching.warp_nocheck(f2, fs)), 4 * b * H * W * (2 * C + 2))
        add(f"l2norm channels      {lvl} C={C} {H}x{W}", timeit(lambda: matching.l2_normalize_channels(f2)), 4 * b * H * W * 2 * C)
        f16 = f2.half().contiguous(memory_format=torch.channels_last)
        add(f"l2norm nhwc16->nchw  {lvl} C={C} {H}x{W}", timeit(lambda: matching.l2_normalize_channels(f16)), 6 * b * H * W * C)
        add(f"  cast + copy + norm {lvl} C={C} {H}x{W}", timeit(lambda: matching.l2_normalize_channels(f16.float().contiguous())), 6 * b * H * W * C)
        if lvl == "L1":
            go = torch.randn(b, 9, 9, H, W, generator=g).to(dev)
            add(f"corr9 backward       {lvl} C={C} {H}x{W}", timeit(lambda: correlation.backward(f1, f2, go, 1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1), reps=5), 4 * b * H * W * (4 * C + 81))
    if not args.only or "upcat" in args.only:
        # decode-head fusion front end at the student's (n = 4) and the teacher's (n = 40) sizes: forward, and the gather
        # backward over the concatenated gradient against the library's bilinear backward on its channel slices
        import os
        from refign_amd.upcat import upsample_concat
        sizes = [(135, 240), (68, 120), (34, 60), (17, 30)]
        for n_ in (4, 40):
            toks = [torch.randn(n_, h * w, 256, device=dev).bfloat16().requires_grad_(n_ == 4) for h, w in sizes]
            nbytes = 2 * (n_ * 135 * 240 * 1024 + sum(n_ * h * w * 256 for h, w in sizes))
            with torch.no_grad():
                add(f"upcat fwd n={n_} 4 x 256 -> 135x240x1024 bf16", timeit(lambda: upsample_concat(toks, sizes, sizes[0])), nbytes)
            if n_ == 4:
                out = upsample_concat(toks, sizes, sizes[0])
                go = torch.randn_like(out)
                for mode in ("1", "0"):
                    os.environ["RFN_UPCAT_BWD"] = mode
                    add(f"upcat bwd n={n_} ({'gather kernel' if mode == '1' else 'library on slices'})",
                        timeit(lambda: torch.autograd.grad(out, toks, go, retain_graph=True), reps=10), nbytes)
                os.environ["RFN_UPCAT_BWD"] = "1"
    if not args.only or "dw" in args.only:
        from refign_amd.dwconv import dwconv3x3_nhwc
        for (B_, H, W, C, dil, dt) in [(40, 135, 240, 1024, 6, torch.bfloat16), (40, 34, 60, 1280, 1, torch.bfloat16),
                                       (40, 135, 240, 256, 1, torch.bfloat16), (4, 135, 240, 256, 1, torch.bfloat16),
                                       (4, 34, 60, 1280, 1, torch.bfloat16), (4, 135, 240, 1024, 12, torch.float32)]:
            x = torch.randn(B_, H, W, C, device=dev).to(dt)
            w = torch.randn(C, 1, 3, 3, device=dev)
            bb = torch.randn(C, device=dev)
            es = x.element_size()
            with torch.no_grad():
                add(f"dwconv fwd {B_}x{H}x{W}x{C} d{dil} {str(dt)[6:]}", timeit(lambda: dwconv3x3_nhwc(x, w, bb, dil)), 2 * x.numel() * es)
            xg = x.clone().requires_grad_()
            wg = w.clone().requires_grad_()
            y = dwconv3x3_nhwc(xg, wg, bb, dil)
            gy = torch.randn_like(y)
            add(f"dwconv fwd+bwd {B_}x{H}x{W}x{C} d{dil} {str(dt)[6:]}", timeit(lambda: torch.autograd.grad(dwconv3x3_nhwc(xg, wg, bb, dil), (xg, wg), gy), reps=5), 7 * x.numel() * es)
    if not args.only or "dw" in args.only:
        from refign_amd.dwconv import dwconv3x3_gelu_tokens
        for (B_, H, W, C) in [(40, 34, 60, 1280), (40, 68, 120, 512), (40, 135, 240, 256), (4, 34, 60, 1280)]:
            x = torch.randn(B_, H * W, C, device=dev).to(torch.bfloat16)
            w = torch.randn(C, 1, 3, 3, device=dev)
            bb = torch.randn(C, device=dev)
            with torch.no_grad():
                add(f"dwconv+gelu fwd (no z) {B_}x{H}x{W}x{C} bf16", timeit(lambda: dwconv3x3_gelu_tokens(x, w, bb, H, W)),
                    2 * x.numel() * 2)
            xg = x.clone().requires_grad_()
            add(f"dwconv+gelu fwd (a, z) {B_}x{H}x{W}x{C} bf16", timeit(lambda: dwconv3x3_gelu_tokens(xg, w, bb, H, W, True)),
                3 * x.numel() * 2)
    if not args.only or "unc" in args.only:
        from refign_amd import align as A
        um = A.UncertaintyModule(1, search_size=9, feed_in_previous=True).to(dev).eval()
        quick = "uncL1x" in args.only                      # fused kernel at level 1 only (tools/unc_ablate.sh)
        for (lvl, H, W) in [("L1", 270, 480), ("L2", 135, 240), ("L3", 32, 32)][:1 if quick else 3]:
            corr = torch.rand(b, 81, H, W, generator=g).to(dev)
            npx = b * H * W
            fl = 2.0 * npx * (49 * 32 * 9 + 25 * 32 * 288 + 9 * 16 * 288 + 6 * 144)
            with torch.no_grad():
                add(f"uncertainty9 front end fused {lvl} {H}x{W}", timeit(lambda: um.patch_statistics(corr), reps=5), 4 * npx * 87, fl)
                if quick:
                    continue
                os.environ["RFN_UNCERT_FUSED"] = "0"
                add(f"uncertainty9 front end library chain {lvl} {H}x{W}", timeit(lambda: um.patch_statistics(corr), reps=3), 4 * npx * 87, fl)
                del os.environ["RFN_UNCERT_FUSED"]
    if not args.only or "ln" in args.only:
        from refign_amd.layernorm import layer_norm
        for (nrow, C) in [(129600, 64), (32640, 128), (8160, 320), (2040, 512), (8160, 1024)]:
            x = torch.randn(nrow, C, device=dev).bfloat16().requires_grad_()
            w = torch.ones(C, device=dev, requires_grad=True)
            bb = torch.zeros(C, device=dev, requires_grad=True)
            with torch.no_grad():
                add(f"layernorm fwd {nrow}x{C} bf16", timeit(lambda: layer_norm(x, w, bb, 1e-6, torch.bfloat16)), 4 * nrow * C)
            y = layer_norm(x, w, bb, 1e-6, torch.bfloat16)
            gy = torch.randn_like(y)
            add(f"layernorm bwd {nrow}x{C} bf16", timeit(lambda: torch.autograd.grad(y, (x, w, bb), gy, retain_graph=True)), 6 * nrow * C)
    if not args.only or "sum" in args.only:
        from refign_amd.params import sum_rows
        for (S, n, dt) in [(8160, 320, torch.bfloat16), (8160, 1280, torch.bfloat16), (32640, 128, torch.bfloat16),
                           (32640, 512, torch.bfloat16), (129600, 64, torch.bfloat16), (129600, 256, torch.bfloat16),
                           (16, 320 * 1280, torch.bfloat16), (64, 64 * 256, torch.bfloat16)]:
            x = torch.randn(S, n, device=dev).to(dt)
            out = torch.zeros(n, device=dev)
            nb = x.numel() * x.element_size() + 8 * n
            add(f"sum_rows (+= into grad) {S}x{n} {str(dt)[6:]}", timeit(lambda: sum_rows(x, out=out, accumulate=True)), nb)
            add(f"  torch: out += x.sum(0) {S}x{n}", timeit(lambda: out.add_(x.sum(0, dtype=torch.float32))), nb)
    if not args.only or "tail" in args.only:
        H, W = 1080, 1920
        lt = (3 * torch.randn(b, 19, H, W, generator=g)).to(dev)
        lr = (3 * torch.randn(b, 19, H, W, generator=g)).to(dev)
        fq = (5 * torch.randn(b, 2, H // 4, W // 4, generator=g)).to(dev)
        lq = (2 * torch.randn(b, 1, H // 4, W // 4, generator=g)).to(dev)
        ff = (5 * torch.randn(b, 2, H, W, generator=g)).to(dev)
        add("warp logits 19x1080x1920 (+mask)", timeit(lambda: matching.warp_nocheck(lr, ff, True)), 4 * b * H * W * (2 * 19 + 2) + b * H * W)
        add("align tail (upsample+cert+warp) 19x1080x1920", timeit(lambda: matching.align_tail(lr, fq, lq)), 4 * b * H * W * (2 * 19 + 1) + b * H * W)
        w, m, c = matching.align_tail(lr, fq, lq)
        add("refine 19x1080x1920", timeit(lambda: refine_mod.refine(lt, w, m, c)), 4 * b * H * W * (3 * 19 + 1) + b * H * W)
        s4 = torch.nn.functional.normalize(torch.randn(b, 512, 16, 16, generator=g), dim=1).to(dev)
        t4 = torch.nn.functional.normalize(torch.randn(b, 512, 16, 16, generator=g), dim=1).to(dev)
        gl = GlobalFeatureCorrelationLayer()
        add("global corr layer 512x16x16", timeit(lambda: gl(s4, t4)), 4 * b * (2 * 512 * 256 + 256 * 256), 2.0 * 512 * 256 * 256 * b)


if __name__ == "__main__":
    main()
