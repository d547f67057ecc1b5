import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from refign_amd.upcat import upsample_concat
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from kbench import timeit
dev = torch.device("cuda:0")
for n in (44, 4):
    sizes = [(135, 240), (68, 120), (34, 60), (17, 30)]
    toks = [torch.randn(n, h * w, 256, device=dev).bfloat16() for h, w in sizes]
    def unfused():
        parts = []
        for t, (h, w) in zip(toks, sizes):
            m = t.transpose(1, 2).reshape(n, 256, h, w)
            parts.append(m if (h, w) == sizes[0] else F.interpolate(m, size=sizes[0], mode='bilinear', align_corners=False))
        return torch.cat(parts, 1)
    with torch.no_grad():
        print(n, "fused  ", timeit(lambda: upsample_concat(toks, sizes, sizes[0]), reps=10), "us")
        print(n, "unfused", timeit(unfused, reps=10), "us")
    tg = [t.clone().requires_grad_() for t in toks]
    go = torch.randn(n, 1024, 135, 240, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    def fb(fused):
        if fused:
            y = upsample_concat(tg, sizes, sizes[0])
        else:
            parts = []
            for t, (h, w) in zip(tg, sizes):
                m = t.transpose(1, 2).reshape(n, 256, h, w)
                parts.append(m if (h, w) == sizes[0] else F.interpolate(m, size=sizes[0], mode='bilinear', align_corners=False))
            y = torch.cat(parts, 1)
        torch.autograd.grad(y, tg, go)
    print(n, "fused fwd+bwd  ", timeit(lambda: fb(True), reps=5), "us")
    print(n, "unfused fwd+bwd", timeit(lambda: fb(False), reps=5), "us")
