"""N4 on the device: the two kernels of csrc/datastep.hip through the C ABI, and the samplers end to end against the goldens
captured from the reference's sampling code (same seed -> the same batches, bit for bit: the arithmetic is u8 / 255, (x - mean) /
std in fp32 with true divisions, as torchvision's ConvertImageDtype + Normalize compute it)."""
import random

import numpy as np
import pytest
import torch
from conftest import golden
from fill import hashed_uniform
from test_datastep_cpu import host_hists, rcs_sampler

pytestmark = pytest.mark.gpu
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def normalized(u8):
    x = torch.from_numpy(np.ascontiguousarray(u8)).to(torch.float32) / 255
    return (x - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)


def test_label_histograms_of_candidate_boxes(dev):
    from refign_amd.datastep import device_label_hists
    rng = np.random.default_rng(0)
    lbl = rng.integers(0, 19, (1024, 2048), dtype=np.uint8)
    lbl[rng.random((1024, 2048)) < 0.05] = 255
    lbl[100:400, 300:1500] = 7                                   # a dominant class: every lane of a wave hits one bin
    boxes = [(0, 0, 1024, 1024), (0, 1024, 1024, 1024), (17, 333, 1000, 1024), (1023, 2047, 1, 1), (5, 7, 3, 2041)] + \
            [(int(rng.integers(0, 512)), int(rng.integers(0, 1024)), 512, 1024) for _ in range(11)]
    got = device_label_hists(torch.from_numpy(lbl).to(dev), boxes)
    np.testing.assert_array_equal(got, host_hists(lbl, boxes))


@pytest.mark.parametrize("flip", [False, True])
def test_crop_flip_normalize_kernel(dev, flip):
    from refign_amd.datastep import crop_flip_normalize
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (3, 97, 263), dtype=np.uint8)
    lbl = rng.integers(0, 256, (97, 263), dtype=np.uint8)
    top, left, h, w = 13, 5, 61, 257
    a, b = crop_flip_normalize(torch.from_numpy(img).to(dev), torch.from_numpy(lbl).to(dev), top, left, h, w, flip)
    ci, cl = img[:, top:top + h, left:left + w], lbl[top:top + h, left:left + w]
    if flip:
        ci, cl = ci[..., ::-1], cl[..., ::-1]
    assert torch.equal(a.cpu(), normalized(ci))                 # bit for bit
    assert b.dtype == torch.int64 and torch.equal(b.cpu(), torch.from_numpy(np.ascontiguousarray(cl)).long())


def test_rare_class_sampler_on_device_matches_reference(dev):
    g = golden("data_rcs")
    s, _, _ = rcs_sampler(g, dev, False)                         # counts from the device kernel
    random.seed(int(g["seed"]))
    for k in range(len(g["images"])):
        img, lbl = s.sample()
        assert torch.equal(img.cpu(), normalized(g["images"][k])), k
        assert torch.equal(lbl.cpu(), torch.from_numpy(g["labels"][k]).long()), k
    assert [random.random() for _ in range(4)] == list(g["random_tail"])


def test_batch_assembler_fills_slots(dev):
    from refign_amd.datastep import PairSampler, UDABatchAssembler
    g, gp = golden("data_rcs"), golden("data_pairs")
    src, _, _ = rcs_sampler(g, dev, False)
    H, W, th, tw = [int(v) for v in gp["size"]]

    def load_pair(i):
        return (torch.from_numpy((hashed_uniform((3, H, W), f"pair/img{i}") * 256).astype(np.uint8)).pin_memory(),
                torch.from_numpy((hashed_uniform((3, H, W), f"pair/ref{i}") * 256).astype(np.uint8)).pin_memory())

    asm = UDABatchAssembler(src, PairSampler(load_pair, (th, tw), dev), 2, dev)
    random.seed(int(g["seed"]))
    batch, ev = asm.assemble([0, 1])
    ev.synchronize()
    assert tuple(batch["image_src"].shape) == (2, 3, *src.size) and batch["semantic_src"].dtype == torch.int64
    for k in range(2):                                           # the source loader's samples come first, in order
        assert torch.equal(batch["image_src"][k].cpu(), normalized(g["images"][k]))
        assert torch.equal(batch["semantic_src"][k].cpu(), torch.from_numpy(g["labels"][k]).long())
    random.seed(int(gp["seed"]))                                 # the pair pipeline alone, from its golden's seed
    ps = PairSampler(load_pair, (th, tw), dev)
    for i in range(3):
        a, b = ps.sample(i)
        assert torch.equal(a.cpu(), normalized(gp["images"][i])) and torch.equal(b.cpu(), normalized(gp["refs"][i]))
    b2, ev2 = asm.assemble([2, 3])
    ev2.synchronize()
    assert b2["image_src"].data_ptr() != batch["image_src"].data_ptr()      # two buffer sets: batch n + 1 beside batch n
