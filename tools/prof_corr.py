#!/usr/bin/env python3
"""tools/prof_corr.py -- run ONE kernel of the hot path a few times (for rocprofv3 --kernel-trace / --pmc).
usage: prof_corr.py [corr_l1|corr_l1_fused|corr_l1_warp|corr_l1_bwd|corr_l2_fused|uncert_l1|tail|refine] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refign_amd import correlation, matching, refine as refine_mod  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "corr_l1_fused"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
b = 2


def feats(C, H, W):
    return (torch.nn.functional.normalize(torch.randn(b, C, H, W, generator=g), dim=1).to(dev),
            torch.nn.functional.normalize(torch.randn(b, C, H, W, generator=g), dim=1).to(dev),
            (5 * torch.randn(b, 2, H, W, generator=g)).to(dev))


if what.startswith("corr_l1"):
    f1, f2, fl = feats(128, 270, 480)
elif what.startswith("corr_l2"):
    f1, f2, fl = feats(256, 135, 240)
if what == "uncert_l1":
    from refign_amd import align as A
    um = A.UncertaintyModule(1, search_size=9, feed_in_previous=True).to(dev).eval()
    corr = torch.rand(b, 81, 270, 480, generator=g).to(dev)

    def fn():
        with torch.no_grad():
            return um.patch_statistics(corr)
elif what == "corr_l1_bwd":
    go = torch.randn(b, 9, 9, 270, 480, generator=g).to(dev)
    fn = lambda: correlation.backward(f1, f2, go, 1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)  # noqa: E731
elif what in ("corr_l1", "corr_l2"):
    fn = lambda: correlation.forward(f1, f2, 1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)  # noqa: E731
elif what.endswith("_fused"):
    fn = lambda: correlation.local_correlation_layer(f2, f1)  # noqa: E731
elif what.endswith("_warp"):
    fn = lambda: correlation.local_correlation_layer(f2, f1, flow=fl)  # noqa: E731
else:
    H, W = 1080, 1920
    lt = (3 * torch.randn(b, 19, H, W, generator=g)).to(dev)
    lr = (3 * torch.randn(b, 19, H, W, generator=g)).to(dev)
    fq = (5 * torch.randn(b, 2, H // 4, W // 4, generator=g)).to(dev)
    lq = (2 * torch.randn(b, 1, H // 4, W // 4, generator=g)).to(dev)
    if what == "tail":
        fn = lambda: matching.align_tail(lr, fq, lq)  # noqa: E731
    else:
        w, m, c = matching.align_tail(lr, fq, lq)
        fn = lambda: refine_mod.refine(lt, w, m, c)  # noqa: E731
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print("done", what, reps)
