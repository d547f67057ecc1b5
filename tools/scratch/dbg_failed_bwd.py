import os, sys, random, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_step_gpu as T
from refign_amd.trainer import Trainer
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "fail"
os.environ["RFN_GRAPH_STUDENT"] = "1"
model = T.build(True, dev)
trainer = Trainer(model, fused_optimizer=False)
sp, mp = model._graphs["source_pass"], model._graphs["mixed_pass"]
if mode == "fail":
    real_bwd = sp.bwd_fn
    def bwd(held, *tensors):
        if torch.cuda.is_current_stream_capturing():
            float(tensors[0].float().sum())
        return real_bwd(held, *tensors)
    sp.bwd_fn = bwd
if mode == "srceager":
    sp.warmup = 10 ** 9
if os.environ.get("DBG_TORCH_SPLIT") == "1":
    from refign_amd import split32
    def split3_torch(x2, order, Kp, stack=False):
        hi, lo = split32.split2(x2)
        rows, K = x2.shape
        terms = (hi, hi, lo) if order == "hhl" else (hi, lo, hi)
        if stack:
            out = torch.zeros((3 * rows, Kp), dtype=torch.bfloat16, device=x2.device)
            for i, t in enumerate(terms):
                out[i * rows:(i + 1) * rows, :K] = t
        else:
            out = torch.zeros((rows, 3 * Kp), dtype=torch.bfloat16, device=x2.device)
            for i, t in enumerate(terms):
                out[:, i * Kp:i * Kp + K] = t
        return out
    split32.split3 = split3_torch
if os.environ.get("DBG_SDPA") == "1":
    from refign_amd import split32
    split32.attention = lambda *a, **k: None


def scan(tag):
    torch.cuda.synchronize()
    gb = model._grad_buffer
    bn = [n for n, b in model.named_buffers() if b.dtype.is_floating_point and bool(torch.isnan(b).any())]
    print(f"   [{tag}] nan: flat {bool(torch.isnan(gb.flat).any())} flat2 {None if gb.flat2 is None else bool(torch.isnan(gb.flat2).any())} "
          f"params {any(bool(torch.isnan(p).any()) for p in model.live_parameters())} buffers {len(bn)} of {sum(1 for n, b in model.named_buffers() if b.dtype.is_floating_point)}: {[n for n in bn if not n.startswith('m_')]}")
real_mf = mp.forward
def mf(*tensors, **kw):
    print("   mixed forward inputs nan:", [bool(torch.isnan(t.float()).any()) for t in tensors], [tuple(t.shape) for t in tensors], tensors[1].tolist())
    scan("before mixed fwd")
    r = real_mf(*tensors, **kw)
    scan("after mixed fwd")
    return r
mp.forward = mf
real_mb = mp.backward
def mb(*tensors):
    st = mp._held[0]
    print("   mixed backward inputs nan:", [bool(torch.isnan(t.float()).any()) for t in tensors], "state", None if st is None else (st["graph"] is not None, st.get("graph_bwd") is not None))
    out = real_mb(*tensors)
    print("   mixed loss", [float(o) for o in out])
    scan("after mixed bwd")
    return out
mp.backward = mb
random.seed(41); np.random.seed(41); torch.manual_seed(41)
for it in range(6):
    batch = T.make_batch(2, 128, 128, 64, dev)
    batch["image_src"] = batch["image_src"] + 0.1 * it
    ac = torch.autocast("cuda", dtype=torch.bfloat16, enabled=os.environ.get("DBG_BF16") == "1")
    with warnings.catch_warnings(), ac:
        warnings.simplefilter("ignore")
        trainer.step(batch, it)
    print(it, [float(model.logged[k]) for k in ("train_loss_src", "train_loss_featdist_src", "train_loss_uda_trg")],
          "params nan", any(bool(torch.isnan(p).any()) for p in model.live_parameters()),
          "grad2", None if getattr(model, "_grad_buffer", None) is None else "buf", "on_second", getattr(model, "_mixed_on_second", None), flush=True)
