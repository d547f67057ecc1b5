"""N4 host logic (refign_amd/datastep.py) against goldens captured from the reference's own sampling code
(tests/golden/make_golden_data.py: Cityscapes.get_rare_class_sample over RandomCrop(cat_max_ratio) + RandomHorizontalFlip, the
(image, image_ref) pipeline, CombinedDataModule.on_before_batch_transfer): with a host label counter injected in place of the
device kernel, the same seed must give the same crops AND leave python's `random` stream where the reference leaves it."""
import random

import numpy as np
import torch
from conftest import golden
from fill import hashed_uniform


def synthetic_set(n, H, W):
    """tests/golden/make_golden_data.py::synthetic_set (closed-form data, not stored in the fixture)"""
    imgs, lbls = [], []
    for i in range(n):
        coarse = (hashed_uniform((H // 32, W // 32), f"data/lbl{i}") * 6).astype(np.uint8)
        coarse[hashed_uniform((H // 32, W // 32), f"data/big{i}") < 0.55] = i % 3
        lbl = np.repeat(np.repeat(coarse, 32, 0), 32, 1)
        u = hashed_uniform((4, 3), f"data/rare{i}")
        for k in range(4):
            y, x = int(u[k, 0] * (H - 24)), int(u[k, 1] * (W - 24))
            lbl[y:y + 12 + 4 * k, x:x + 20] = 11 + k
        lbl[hashed_uniform((H, W), f"data/ign{i}") < 0.03] = 255
        imgs.append((hashed_uniform((3, H, W), f"data/img{i}") * 256).astype(np.uint8))
        lbls.append(lbl)
    return imgs, lbls


def host_hists(label, boxes):
    return np.stack([np.bincount(label[t:t + h, l:l + w].reshape(-1), minlength=256) for (t, l, h, w) in boxes]).astype(np.int64)


def rcs_sampler(g, device, hists_from_host):
    from refign_amd.datastep import RareClassSourceSampler
    H, W, th, tw = [int(v) for v in g["size"]]
    imgs, lbls = synthetic_set(5, H, W)
    classes = [int(c) for c in g["classes"]]
    state = {}

    def load(index):
        state["lbl"] = lbls[index]
        return torch.from_numpy(imgs[index].copy()), torch.from_numpy(lbls[index].copy())

    s = RareClassSourceSampler(load, classes, g["prob"], {c: [i for i in range(5) if (lbls[i] == c).sum() > 0] for c in classes},
                               (th, tw), device, cat_max_ratio=float(g["cat_max_ratio"]), rcs_min_pixels=int(g["min_pixels"]),
                               rcs_min_crop_ratio=float(g["min_crop_ratio"]),
                               hists=(lambda boxes: host_hists(state["lbl"], boxes)) if hists_from_host else None)
    return s, imgs, lbls


def test_rare_class_sampling_draws_match_reference():
    g = golden("data_rcs")
    s, imgs, lbls = rcs_sampler(g, "cpu", True)
    random.seed(int(g["seed"]))
    for k in range(len(g["images"])):
        index, _, _, (top, left, h, w), flip = s.draw()
        img = imgs[index][:, top:top + h, left:left + w]
        lbl = lbls[index][top:top + h, left:left + w]
        if flip:
            img, lbl = img[..., ::-1], lbl[..., ::-1]
        np.testing.assert_array_equal(img, g["images"][k], err_msg=f"sample {k}")
        np.testing.assert_array_equal(lbl, g["labels"][k], err_msg=f"sample {k}")
    assert [random.random() for _ in range(4)] == list(g["random_tail"])      # the stream stands where the reference's stands


def test_pair_crop_draws_match_reference():
    from refign_amd.datastep import draw_crop
    g = golden("data_pairs")
    H, W, th, tw = [int(v) for v in g["size"]]
    random.seed(int(g["seed"]))
    for i in range(len(g["images"])):
        img = (hashed_uniform((3, H, W), f"pair/img{i}") * 256).astype(np.uint8)
        ref = (hashed_uniform((3, H, W), f"pair/ref{i}") * 256).astype(np.uint8)
        (top, left, h, w), _ = draw_crop(H, W, (th, tw))
        flip = random.random() < 0.5
        a, b = img[:, top:top + h, left:left + w], ref[:, top:top + h, left:left + w]
        if flip:
            a, b = a[..., ::-1], b[..., ::-1]
        np.testing.assert_array_equal(a, g["images"][i])
        np.testing.assert_array_equal(b, g["refs"][i])
    full = (hashed_uniform((3, th, tw), "pair/full") * 256).astype(np.uint8)
    (top, left, h, w), _ = draw_crop(th, tw, (th, tw))                             # image of the crop's size: no draw
    assert (top, left, h, w) == (0, 0, th, tw)
    flip = random.random() < 0.5
    np.testing.assert_array_equal(full[..., ::-1] if flip else full, g["full"])
    assert [random.random() for _ in range(4)] == list(g["random_tail"])


def test_merge_matches_reference():
    from refign_amd.datastep import merge_batches
    g = golden("data_merge")
    sub = [{k[4:]: torch.from_numpy(g[k]) for k in g if k.startswith("in0_")},
           {k[4:]: torch.from_numpy(g[k]) for k in g if k.startswith("in1_")}]
    out = merge_batches(sub)
    assert sorted(out) == sorted(k[4:] for k in g if k.startswith("out_"))
    for k, v in out.items():
        np.testing.assert_array_equal(v.numpy(), g["out_" + k])
