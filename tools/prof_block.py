import sys, torch
sys.path.insert(0, "/root/repo")
from refign_amd import seg
from refign_amd.trainer import FlatGradBuffer
dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = seg.Block(320, 5, 4, True, drop_path=0.1, sr_ratio=2, norm_layer=lambda d: seg.LayerNorm(d, eps=1e-6)).to(dev).train()
FlatGradBuffer(list(blk.parameters()))
B, H, W = 4, 34, 60
x = torch.randn(B, H * W, 320, device=dev, dtype=torch.bfloat16, requires_grad=True)
m32 = torch.tensor([[1.1, 0.0, 1.1, 1.1], [1.1, 1.1, 0.0, 1.1]], device=dev)
m16 = m32.to(torch.bfloat16).view(2, B, 1, 1)
def run():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = blk(x, H, W, m16, m32)
    y.backward(torch.ones_like(y))
for _ in range(3): run()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run()
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0]
rows.sort(key=lambda r: -r[2])
for k, c, t in rows[:60]:
    print(f"{t:8.1f} us  x{c:3d}  {k[:110]}")
