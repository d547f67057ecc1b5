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
