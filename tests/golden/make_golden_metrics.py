"""tests/golden/make_golden_metrics.py -- G16: the reference's SparseEPE metric (helpers/metrics.py:35-262) on synthetic
flows and correspondences.  torchmetrics is not installed: its `Metric` base is a throw-away stub whose add_state() simply
creates the attribute, which is all the reference's update() / compute() need on one process.
    python tests/golden/make_golden_metrics.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_import as R  # noqa: E402

R.setup()
sys.modules["torchmetrics"].Metric.add_state = lambda self, name, default, dist_reduce_fx=None: setattr(self, name, default.clone())
from fill import hashed_uniform  # noqa: E402
from make_golden_modules import save, t  # noqa: E402


def main():
    hm = R.ref_module("helpers.metrics")
    B, H, W, n = 2, 60, 90, 400
    flow = ((hashed_uniform((B, 2, H, W), "g16/flow") - 0.5) * 12).astype(np.float32)
    unc = hashed_uniform((B, 1, H, W), "g16/unc").astype(np.float32)
    pts_t = [np.stack([hashed_uniform((n,), f"g16/xt{b}") * (W + 8) - 4, hashed_uniform((n,), f"g16/yt{b}") * (H + 8) - 4], 1).astype(np.float32)
             for b in range(B)]
    pts_s = []
    for b in range(B):
        ix = np.clip(np.round(pts_t[b][:, 0]), 0, W - 1).astype(int)
        iy = np.clip(np.round(pts_t[b][:, 1]), 0, H - 1).astype(int)
        # ground truth = the estimate plus an error that grows with the predicted uncertainty (so AUSE is informative)
        err = (hashed_uniform((n, 2), f"g16/err{b}") - 0.5) * 14 * (0.2 + unc[b, 0, iy, ix])[:, None]
        pts_s.append((pts_t[b] + flow[b][:, iy, ix].T + err).astype(np.float32))
    m = hm.SparseEPE(uncertainty_estimation=True)
    m.update(t(flow), [t(p) for p in pts_s], [t(p) for p in pts_t], (H, W), t(unc))
    out = {k: float(v) for k, v in m.compute().items()}
    m2 = hm.SparseEPE(uncertainty_estimation=False)
    m2.update(t(flow[:1]), [t(pts_s[0])], [t(pts_t[0])], (H, W))
    out1 = {k + "_first": float(v) for k, v in m2.compute().items()}
    save("metric_sparse_epe", flow=flow, unc=unc, pts_s=np.stack(pts_s), pts_t=np.stack(pts_t),
         nbr_valid_corr=np.int64(int(m.nbr_valid_corr)), **{k: np.float64(v) for k, v in {**out, **out1}.items()})
    print(out, out1, int(m.nbr_valid_corr))


if __name__ == "__main__":
    main()
