import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda:0")
wl = bench.RefignStep(dev, 2, 1234, adapt_to_ref="--plain" not in sys.argv)
m = wl.model
for i in range(10):
    wl.step()
    torch.cuda.synchronize()
    g = m._graphs
    print(i, "conc", m.__dict__.get("_mixed_concurrent_steps"), "early", m.__dict__.get("_mixed_early_forwards"),
          "on_second", getattr(m, "_mixed_on_second", None), "src cap", g["source_pass"].captured(), "mix cap", g["mixed_pass"].captured(),
          "probe", getattr(m, "_mix_stream_probe", None), "heads", m.__dict__.get("_adapted_to_ref_steps"), "src states", [(k[-1] if isinstance(k, tuple) else k, st["graph"] is not None, st.get("graph_bwd") is not None, st["failed"]) for k, st in g["source_pass"].states.items()],
          "mix states", [(st["graph"] is not None, st.get("graph_bwd") is not None, st["failed"]) for k, st in g["mixed_pass"].states.items()], flush=True)
