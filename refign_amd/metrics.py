"""Evaluation metrics of the segmentation path: host mirror of helpers/metrics.py's `IoU`
(a torchmetrics.JaccardIndex with ignore_index handled in `update`, helpers/metrics.py:264-387) and of its metric
collection, without torchmetrics: the state is one (C, C) confusion matrix on the device (rows = target, columns =
prediction, like torchmetrics' `_confusion_matrix_update`), summed over ranks at `compute()` (`dist_reduce_fx="sum"`).

Constructor keywords are the reference's / JaccardIndex 0.9's: num_classes, ignore_index, absent_score, threshold,
average ('macro' | 'weighted' | 'none' | None), over_present_classes, compute_on_step (accepted, unused)."""
import torch
import torch.distributed as dist
import torch.nn as nn


class IoU(nn.Module):
    def __init__(self, num_classes, ignore_index=None, absent_score=0.0, threshold=0.5, average='macro',
                 over_present_classes=False, multilabel=False, compute_on_step=None, **kwargs):
        super().__init__()
        if average not in ('macro', 'weighted', 'none', None):
            raise ValueError(f"The `average` has to be one of ['macro', 'weighted', 'none', None], got {average}.")
        if multilabel:
            raise NotImplementedError("multilabel IoU is not used by the reference's configs")
        self.num_classes, self.ignore_index, self.absent_score = num_classes, ignore_index, absent_score
        self.threshold, self.average, self.over_present_classes = threshold, average, over_present_classes
        self.register_buffer("confmat", torch.zeros(num_classes, num_classes, dtype=torch.long), persistent=False)

    def reset(self):
        self.confmat.zero_()

    @torch.no_grad()
    def update(self, preds, target):
        """preds: (B, C, H, W) scores (arg-max taken, as torchmetrics does for multi-dimensional multi-class input) or
        (B, H, W) labels; target: (B, H, W) labels; pixels with target == ignore_index do not count."""
        target = target.reshape(-1)
        valid = target != self.ignore_index if self.ignore_index is not None else torch.ones_like(target, dtype=torch.bool)
        if preds.dim() == 4:
            preds = preds.argmax(1)
        preds = preds.reshape(-1)[valid]
        target = target[valid]
        if self.confmat.device != target.device:
            self.confmat = self.confmat.to(target.device)
        idx = target.long() * self.num_classes + preds.long()
        self.confmat += torch.bincount(idx, minlength=self.num_classes ** 2).view(self.num_classes, self.num_classes)

    __call__ = update                                     # compute_on_step=False: a call only accumulates

    def _scores(self, confmat):
        inter = torch.diag(confmat)
        union = confmat.sum(0) + confmat.sum(1) - inter
        scores = inter.float() / union.float()
        scores[union == 0] = self.absent_score
        present = confmat.sum(dim=1) != 0
        return scores[present] if self.over_present_classes else scores

    @torch.no_grad()
    def compute(self):
        confmat = self.confmat.clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(confmat)
        if self.average in ('none', None):
            return self._scores(confmat)
        if self.average == 'macro':
            return self._scores(confmat).mean()
        weights = confmat.sum(dim=1).float() / confmat.sum().float()
        return (weights * self._scores(confmat)).sum() if not self.over_present_classes else \
            (weights[confmat.sum(dim=1) != 0] * self._scores(confmat)).sum()


class MyMetricCollection(nn.ModuleDict):
    """helpers/metrics.py:13-32: a dict of metrics; compute() flattens dict-valued results to 'name_key'."""

    def __init__(self, metrics=None):
        super().__init__(metrics or {})

    def compute(self):
        out = {}
        for name, m in self.items():
            val = m.compute()
            if isinstance(val, dict):
                out.update({f"{name}_{k}": v for k, v in val.items()})
            else:
                out[name] = val
        return out

    def reset(self):
        for m in self.values():
            m.reset()


def build_collections(metrics_cfg, instantiate):
    """segmentation_model.py:93-98: {'val' | 'test': {dataset: [specs]}} -> two collections keyed
    '<split>_<dataset>_<ClassName>'.  Specs of metrics that are not built here (SparseEPE) are skipped."""
    from .config import OutOfScopeError
    out = []
    for split in ('val', 'test'):
        items = {}
        for ds, specs in (metrics_cfg or {}).get(split, {}).items():
            for el in specs:
                try:
                    items[f"{split}_{ds}_{el['class_path'].split('.')[-1]}"] = instantiate(tuple(), el)
                except OutOfScopeError:
                    pass
        out.append(MyMetricCollection(items))
    return out


class SparseEPE(nn.Module):
    """helpers/metrics.py:35-262 without torchmetrics: end-point error of a dense flow at sparse ground-truth
    correspondences (the matcher's evaluation on MegaDepth / RobotCar), PCK at 1 / 3 / 5 / 10 px, and -- with
    `uncertainty_estimation` -- the area under the sparsification-error curve of the predicted confidence (AUSE, 50
    quantile intervals: EPE of the pixels that remain when the most uncertain q % are removed, against the oracle that
    removes the worst q % first).  Seven running sums, summed over ranks at compute() (`dist_reduce_fx="sum"`).

    update(t_s_flow (B,2,h,w), corr_pts_s, corr_pts_t [B x (n,2) pixel coordinates (x, y)], out_size (h,w),
    uncertainty_est (B,1,h,w) or None): the flow is looked up at the ROUNDED target points."""

    def __init__(self, uncertainty_estimation=False, compute_on_step=None, **kwargs):
        super().__init__()
        self.uncertainty_estimation = uncertainty_estimation
        for name in ("AEPE", "PCK_1", "PCK_3", "PCK_5", "PCK_10", "AUSE_AEPE"):
            self.register_buffer(name, torch.zeros((), dtype=torch.double), persistent=False)
        for name in ("nbr_valid_corr", "nbr_samples"):
            self.register_buffer(name, torch.zeros((), dtype=torch.long), persistent=False)

    def reset(self):
        for b in self.buffers():
            b.zero_()

    @torch.no_grad()
    def update(self, t_s_flow, corr_pts_s, corr_pts_t, out_size, uncertainty_est=None):
        h, w = out_size
        assert tuple(t_s_flow.shape[-2:]) == (h, w)
        if self.AEPE.device != t_s_flow.device:
            self.to(t_s_flow.device)
        for b in range(t_s_flow.shape[0]):
            xs, ys, xt, yt = corr_pts_s[b][:, 0], corr_pts_s[b][:, 1], corr_pts_t[b][:, 0], corr_pts_t[b][:, 1]
            ok = (torch.round(xs) >= 0) & (torch.round(xs) < w) & (torch.round(ys) >= 0) & (torch.round(ys) < h) & \
                (torch.round(xt) >= 0) & (torch.round(xt) < w) & (torch.round(yt) >= 0) & (torch.round(yt) < h)
            n = int(ok.sum())
            if n == 0:
                continue
            xs, ys, xt, yt = xs[ok], ys[ok], xt[ok], yt[ok]
            iy, ix = torch.round(yt).long(), torch.round(xt).long()
            gt = torch.stack([xs - xt, ys - yt], dim=1)
            est = torch.stack([t_s_flow[b, 0, iy, ix], t_s_flow[b, 1, iy, ix]], dim=1)
            epe = torch.linalg.norm(gt - est, ord=2, dim=1)
            self.AEPE += epe.mean()
            self.PCK_1 += (epe <= 1.0).sum()
            self.PCK_3 += (epe <= 3.0).sum()
            self.PCK_5 += (epe <= 5.0).sum()
            self.PCK_10 += (epe <= 10.0).sum()
            self.nbr_valid_corr += n
            self.nbr_samples += 1
            if self.uncertainty_estimation:
                self.AUSE_AEPE += self.compute_aucs(gt, est, uncertainty_est[b, 0, iy, ix])['EPE']

    __call__ = update

    @staticmethod
    def compute_aucs(gt, pred, uncert, intervals=50):
        epe = torch.linalg.norm(gt - pred, ord=2, dim=1)
        quants = [t / intervals for t in range(intervals)]
        plotx = torch.tensor([t / intervals for t in range(intervals + 1)], device=gt.device)

        def curve(score):                               # score: high = removed first; keep `score >= its q-quantile` of -score
            keep_first = -score
            vals = [epe[keep_first >= torch.quantile(keep_first.float(), q)].mean() for q in quants]
            return torch.stack(vals + [torch.zeros((), device=gt.device)])
        sparse, oracle = curve(uncert), curve(epe)
        mmax = oracle.max() + 1e-6
        return {'EPE': torch.abs(torch.trapz(sparse / mmax, x=plotx) - torch.trapz(oracle / mmax, x=plotx))}

    @torch.no_grad()
    def compute(self):
        vals = {k: getattr(self, k).clone() for k in ("AEPE", "PCK_1", "PCK_3", "PCK_5", "PCK_10", "AUSE_AEPE",
                                                      "nbr_valid_corr", "nbr_samples")}
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            for v in vals.values():
                dist.all_reduce(v)
        out = {'AEPE': vals["AEPE"] / vals["nbr_samples"].double(),
               'PCK_1': vals["PCK_1"] / vals["nbr_valid_corr"].double(),
               'PCK_3': vals["PCK_3"] / vals["nbr_valid_corr"].double(),
               'PCK_5': vals["PCK_5"] / vals["nbr_valid_corr"].double(),
               'PCK_10': vals["PCK_10"] / vals["nbr_valid_corr"].double()}
        if self.uncertainty_estimation:
            out['AUSE_AEPE'] = vals["AUSE_AEPE"] / vals["nbr_samples"].double()
        return out
