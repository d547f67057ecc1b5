"""CPU: refign_amd/metrics.py -- IoU with ignore_index against a brute-force count, its
averaging modes (helpers/metrics.py:303-387), the cross-rank sum over gloo, and the metric collections a reference
config declares."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def brute(pred, tgt, C, ignore):
    inter, union, support = np.zeros(C), np.zeros(C), np.zeros(C)
    keep = tgt != ignore
    p, t = pred[keep], tgt[keep]
    for c in range(C):
        inter[c] = np.sum((p == c) & (t == c))
        union[c] = np.sum((p == c) | (t == c))
        support[c] = np.sum(t == c)
    return inter, union, support


def test_iou_matches_brute_force_and_reference_averaging():
    from refign_amd.metrics import IoU
    rng = np.random.RandomState(0)
    C = 7
    tgt = rng.randint(0, 5, size=(3, 20, 30))            # classes 5, 6 never in the target
    tgt[rng.rand(3, 20, 30) < 0.1] = 255
    logits = rng.randn(3, C, 20, 30).astype(np.float32)
    logits[:, 6] = -10                                    # class 6 never predicted either: absent everywhere
    pred = logits.argmax(1)
    inter, union, support = brute(pred, tgt, C, 255)
    iou = np.where(union > 0, inter / np.maximum(union, 1), 0.25)
    for from_logits in (True, False):
        m = IoU(num_classes=C, ignore_index=255, absent_score=0.25, average='none')
        for i in range(3):                                # accumulates over calls
            m(torch.from_numpy(logits[i:i + 1]) if from_logits else torch.from_numpy(pred[i:i + 1]),
              torch.from_numpy(tgt[i:i + 1]))
        np.testing.assert_allclose(m.compute().numpy(), iou, rtol=1e-6)
    a = (torch.from_numpy(logits), torch.from_numpy(tgt))
    mk = lambda **kw: IoU(num_classes=C, ignore_index=255, absent_score=0.25, **kw)  # noqa: E731
    m = mk(); m(*a)
    assert abs(float(m.compute()) - iou.mean()) < 1e-6                                   # macro over ALL classes
    m = mk(over_present_classes=True); m(*a)
    assert abs(float(m.compute()) - iou[support > 0].mean()) < 1e-6                      # ... over the present ones
    m = mk(average='weighted'); m(*a)
    assert abs(float(m.compute()) - (support / support.sum() * iou).sum()) < 1e-6
    m.reset()
    assert int(m.confmat.sum()) == 0
    with pytest.raises(ValueError):
        mk(average='micro')


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from refign_amd.metrics import IoU
    rng = np.random.RandomState(5)
    tgt, pred = rng.randint(0, 4, size=(4, 8, 8)), rng.randint(0, 4, size=(4, 8, 8))
    m = IoU(num_classes=4, ignore_index=255)
    m(torch.from_numpy(pred[rank * 2:rank * 2 + 2]), torch.from_numpy(tgt[rank * 2:rank * 2 + 2]))
    full = IoU(num_classes=4, ignore_index=255)
    full.update(torch.from_numpy(pred), torch.from_numpy(tgt))
    inter = torch.diag(full.confmat)
    want = (inter.float() / (full.confmat.sum(0) + full.confmat.sum(1) - inter).float()).mean()
    torch.save((float(m.compute()), float(want)), f"{out}/iou_{rank}.pt")
    dist.destroy_process_group()


def test_iou_sums_confusion_matrices_over_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        got, want = torch.load(f"{tmp_path}/iou_{rank}.pt")
        assert abs(got - want) < 1e-6


def test_metric_collections_of_a_reference_config():
    """The `metrics` section of a refign_* YAML, as segmentation_model.py:93-98 turns it into two collections."""
    from refign_amd.metrics import IoU, build_collections
    from refign_amd.config import instantiate_class
    cfg = {"val": {"DarkZurich": [{"class_path": "helpers.metrics.IoU",
                                   "init_args": {"ignore_index": 255, "num_classes": 19, "compute_on_step": False}}]},
           "test": {"DarkZurich": [{"class_path": "helpers.metrics.IoU",
                                    "init_args": {"ignore_index": 255, "num_classes": 19, "compute_on_step": False}}],
                    "RobotCarMatching": [{"class_path": "helpers.metrics.SparseEPE", "init_args": {}}]}}
    val, test = build_collections(cfg, instantiate_class)
    assert list(val.keys()) == ["val_DarkZurich_IoU"]
    assert list(test.keys()) == ["test_DarkZurich_IoU", "test_RobotCarMatching_SparseEPE"]
    assert isinstance(val["val_DarkZurich_IoU"], IoU) and val["val_DarkZurich_IoU"].num_classes == 19
    val["val_DarkZurich_IoU"](torch.zeros(1, 4, 4, dtype=torch.long), torch.zeros(1, 4, 4, dtype=torch.long))
    assert set(val.compute()) == {"val_DarkZurich_IoU"}


def test_sparse_epe_matches_reference():
    """refign_amd.metrics.SparseEPE against values computed by the reference's own class (G16: AEPE, PCK at 1/3/5/10 px,
    AUSE of the sparsification curves, correspondences outside the image dropped) -- accumulated over two samples, and
    the deterministic variant on one."""
    from conftest import golden
    from refign_amd.metrics import SparseEPE
    z = golden("metric_sparse_epe")
    flow, unc = torch.from_numpy(z["flow"]), torch.from_numpy(z["unc"])
    ps, pt = [torch.from_numpy(p) for p in z["pts_s"]], [torch.from_numpy(p) for p in z["pts_t"]]
    m = SparseEPE(uncertainty_estimation=True)
    for b in range(2):                                   # sample by sample == one batched update
        m(flow[b:b + 1], ps[b:b + 1], pt[b:b + 1], tuple(flow.shape[-2:]), unc[b:b + 1])
    assert int(m.nbr_valid_corr) == int(z["nbr_valid_corr"]) and int(m.nbr_samples) == 2
    out = m.compute()
    for k in ("AEPE", "PCK_1", "PCK_3", "PCK_5", "PCK_10", "AUSE_AEPE"):
        assert abs(float(out[k]) - float(z[k])) <= 1e-5 * max(abs(float(z[k])), 1e-3), k
    m2 = SparseEPE()
    m2.update(flow[:1], ps[:1], pt[:1], tuple(flow.shape[-2:]))
    out2 = m2.compute()
    assert "AUSE_AEPE" not in out2
    for k in ("AEPE", "PCK_1", "PCK_10"):
        assert abs(float(out2[k]) - float(z[k + "_first"])) <= 1e-5 * max(abs(float(z[k + "_first"])), 1e-3), k
    m.reset()
    assert int(m.nbr_samples) == 0 and float(m.AEPE) == 0.0
