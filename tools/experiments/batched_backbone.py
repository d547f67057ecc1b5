"""Would ONE student backbone pass over the source AND the mixed batch (8 HRDA views) beat two passes over 4 views each?
Times the graphed backbone segment (forward graph, backward graph) for b = 2 and b = 4 images of 1080 x 1920, each on its own."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from refign_amd.graphs import GraphedSegment, _leaves

dev = torch.device("cuda:0")
wl = bench.RefignStep(dev, 2, 1234)
m = wl.model
m.train()
off = torch.zeros(2, dtype=torch.long, device=dev)
for b in (2, 4):
    seg = GraphedSegment(m._backbone_fn, f"bb{b}")
    x = torch.randn(b, 3, 1080, 1920, device=dev)
    def once():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = seg(x, off)
            leaves = [t for t in _leaves(out, []) if torch.is_tensor(t) and t.requires_grad]
        return leaves
    for _ in range(3):
        leaves = once()
        torch.autograd.backward(leaves, [torch.ones_like(t) for t in leaves])
    assert seg.captured()
    st = next(iter(seg.states.values()))
    torch.cuda.synchronize()
    for name, g in (("fwd", st["graph"]), ("bwd", st["graph_bwd"])):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"b = {b} images ({2 * b} views): backbone {name} {min(ts):.2f} ms", flush=True)
    for p in m.parameters():
        p.grad = None
