"""tests/golden/make_golden_data.py -- N4 goldens: the reference's SAMPLING code run on synthetic in-memory data --
Cityscapes.get_rare_class_sample (data_modules/datasets/cityscapes.py:139-158) over its transform pipeline ToTensor ->
RandomCrop(size, cat_max_ratio) -> RandomHorizontalFlip (data_modules/transforms.py:282-390), the (image, image_ref) pipeline
of the target set, and CombinedDataModule.on_before_batch_transfer (combined_data_module.py:263-310).  cv2 and torchvision are
not installed here: they get throw-away stubs (only `pil_to_tensor`-free paths are used: the samples start as uint8 tensors,
`hflip` is the one torchvision function the called code reaches and is stubbed as flip(-1), its documented behaviour).
ConvertImageDtype / Normalize are torchvision subclasses and are not run (third-party arithmetic: u8 / 255, (x - mean) / std).
    python tests/golden/make_golden_data.py      ->  data_rcs.npz, data_pairs.npz, data_merge.npz"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402
from fill import hashed_uniform  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} kB)")


def stubs():
    class _Base(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
    tv = R._stub("torchvision")
    tt = R._stub("torchvision.transforms", RandomRotation=_Base, ColorJitter=_Base, ConvertImageDtype=_Base, Normalize=_Base,
                 GaussianBlur=_Base, CenterCrop=_Base, Compose=object)
    tf = R._stub("torchvision.transforms.functional", hflip=lambda x: x.flip(-1))
    tv.transforms, tt.functional = tt, tf
    R._stub("cv2")
    sys.modules["pytorch_lightning"].LightningDataModule = object
    R._pkg("data_modules", os.path.join(R.REF, "data_modules"))
    R._pkg("data_modules.datasets", os.path.join(R.REF, "data_modules", "datasets"))


def synthetic_set(n, H, W):
    """label maps with large constant blocks (so that the category-ratio test of RandomCrop fails often), a few small rare-class
    patches (so that rare-class re-draws happen) and ignore pixels; images = hashed bytes"""
    imgs, lbls = [], []
    for i in range(n):
        coarse = (hashed_uniform((H // 32, W // 32), f"data/lbl{i}") * 6).astype(np.uint8)          # classes 0..5, 32 x 32 blocks
        coarse[hashed_uniform((H // 32, W // 32), f"data/big{i}") < 0.55] = i % 3                    # dominant class
        lbl = np.repeat(np.repeat(coarse, 32, 0), 32, 1)
        u = hashed_uniform((4, 3), f"data/rare{i}")
        for k in range(4):                                                                           # rare classes 11..14
            y, x = int(u[k, 0] * (H - 24)), int(u[k, 1] * (W - 24))
            lbl[y:y + 12 + 4 * k, x:x + 20] = 11 + k
        lbl[hashed_uniform((H, W), f"data/ign{i}") < 0.03] = 255
        imgs.append((hashed_uniform((3, H, W), f"data/img{i}") * 256).astype(np.uint8))
        lbls.append(lbl)
    return imgs, lbls


def g_rcs(tr, cs):
    H, W, size = 192, 384, (96, 128)
    imgs, lbls = synthetic_set(5, H, W)
    crop = tr.RandomCrop(size=list(size), cat_max_ratio=0.75)
    flip = tr.RandomHorizontalFlip()

    def pipeline(sample):
        return flip(crop(sample))

    classes = [11, 12, 13, 14, 3]
    prob = np.array([0.3, 0.25, 0.2, 0.15, 0.1])
    ns = types.SimpleNamespace(load_keys=["image", "semantic"], rcs_classes=classes, rcs_classprob=prob,
                               indices_with_class={c: [i for i in range(5) if (lbls[i] == c).sum() > 0] for c in classes},
                               rcs_min_crop_ratio=2.0, rcs_min_pixels=60)
    ns.load_and_augment_sample = lambda index: pipeline({"image": torch.from_numpy(imgs[index].copy()),
                                                         "semantic": torch.from_numpy(lbls[index].copy())})
    random.seed(2024)
    out_img, out_lbl = [], []
    for _ in range(16):
        s = cs.Cityscapes.get_rare_class_sample(ns)
        out_img.append(s["image"].numpy().copy())
        out_lbl.append(s["semantic"].numpy().copy())
    tail = [random.random() for _ in range(4)]                     # where the random stream stands afterwards
    save("data_rcs", size=np.array([H, W, *size]), classes=np.array(classes), prob=prob, min_pixels=np.int64(60),
         min_crop_ratio=np.float64(2.0), cat_max_ratio=np.float64(0.75), seed=np.int64(2024), images=np.stack(out_img),
         labels=np.stack(out_lbl), random_tail=np.array(tail))


def g_pairs(tr):
    H, W, size = 160, 288, (96, 128)
    crop = tr.RandomCrop(size=list(size))
    flip = tr.RandomHorizontalFlip()
    random.seed(77)
    a, b = [], []
    for i in range(8):
        img = (hashed_uniform((3, H, W), f"pair/img{i}") * 256).astype(np.uint8)
        ref = (hashed_uniform((3, H, W), f"pair/ref{i}") * 256).astype(np.uint8)
        s = flip(crop({"image": torch.from_numpy(img), "image_ref": torch.from_numpy(ref)}))
        a.append(s["image"].numpy().copy())
        b.append(s["image_ref"].numpy().copy())
    # full-size case: no draw for the crop (get_params returns early)
    img = (hashed_uniform((3, *size), "pair/full") * 256).astype(np.uint8)
    s = flip(crop({"image": torch.from_numpy(img)}))
    save("data_pairs", size=np.array([H, W, *size]), seed=np.int64(77), images=np.stack(a), refs=np.stack(b),
         full=s["image"].numpy().copy(), random_tail=np.array([random.random() for _ in range(4)]))


def g_merge(cdm):
    ns = types.SimpleNamespace(trainer=types.SimpleNamespace(training=True), ignore_every_second_semantic_training_batch=False)
    g = torch.Generator().manual_seed(5)
    sub = [{"image": torch.randn(2, 3, 4, 5, generator=g), "semantic": torch.randint(0, 19, (2, 4, 5), generator=g)},
           {"image": torch.randn(2, 3, 4, 5, generator=g), "image_ref": torch.randn(2, 3, 4, 5, generator=g)}]
    out = cdm.CombinedDataModule.on_before_batch_transfer(ns, sub, 0)
    save("data_merge", keys=np.array(sorted(out)), **{"in0_" + k: v.numpy() for k, v in sub[0].items()},
         **{"in1_" + k: v.numpy() for k, v in sub[1].items()}, **{"out_" + k: v.numpy() for k, v in out.items()})


if __name__ == "__main__":
    R.setup()
    stubs()
    tr = R.ref_module("data_modules.transforms")
    cs = R.ref_module("data_modules.datasets.cityscapes")
    g_rcs(tr, cs)
    g_pairs(tr)
    # combined_data_module.py star-imports every dataset (h5py, cv2 ...): instead of stubbing all of that, the one method the golden
    # needs is compiled from the reference file where it lies (authoring container only; nothing of it is stored)
    import ast
    path = os.path.join(R.REF, "data_modules", "combined_data_module.py")
    tree = ast.parse(open(path).read())
    fn = next(n for c in tree.body if isinstance(c, ast.ClassDef) and c.name == "CombinedDataModule"
              for n in c.body if isinstance(n, ast.FunctionDef) and n.name == "on_before_batch_transfer")
    env = {"torch": torch, "random": random}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), env)
    g_merge(types.SimpleNamespace(CombinedDataModule=types.SimpleNamespace(on_before_batch_transfer=env["on_before_batch_transfer"])))
