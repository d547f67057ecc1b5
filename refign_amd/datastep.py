"""refign_amd/datastep.py -- N4, second part: SAMPLING of the UDA iteration with the pixel work on the device.

What the reference does on the host for every training sample (data_modules/):
  * source set, `Cityscapes.get_rare_class_sample` (datasets/cityscapes.py:139-158): draw a rare class c and an image that has it,
    run the transform pipeline, and while the crop holds too few pixels of c run the pipeline again (up to 10 times);
  * the pipeline of refign_hrda_star.yaml:10-40: ToTensor, RandomCrop(size, cat_max_ratio=0.75) -- which itself re-draws its box
    up to 10 times while one category covers more than 75 % of the crop (transforms.py:282-361: a `torch.unique` of a 1024 x 1024
    label crop per candidate) -- RandomHorizontalFlip (:363-390), ConvertImageDtype (:438-464), Normalize (:467-495);
  * target set: RandomCrop + flip of (image, image_ref) with one set of parameters;
  * `CombinedDataModule.on_before_batch_transfer` (combined_data_module.py:263-310): torch.cat of the sub-batches into
    {image_src, semantic_src, image_trg, image_ref}.
Here the uint8 image / label map go to the device once (pinned, asynchronous), every candidate box of a RandomCrop call is
counted by ONE kernel launch (csrc/datastep.hip: the draws of a call do not depend on the outcomes, only where the chain stops
does -- the host draws the whole chain, asks once and rewinds python's `random` stream to the stop, so the stream is consumed
exactly as the reference consumes it), and crop + flip + conversion + normalisation write straight into the sample's slot of
the batch tensors.  Same `random` calls in the same order: seeded alike, it yields the reference's batches
(tests/test_datastep_*.py against goldens captured from the reference's own code, tests/golden/make_golden_data.py)."""
import random

import numpy as np
import torch

from . import _lib
from ._tensor import current_stream, on_device, ptr

IMNET_MEAN = (0.485, 0.456, 0.406)
IMNET_STD = (0.229, 0.224, 0.225)


def device_label_hists(label_u8, boxes):
    """(K, 256) int64 numpy: label histograms of the K <= 16 crop boxes (top, left, h, w) of a DEVICE uint8 label map.  One launch,
    one small device-to-host copy (the decision that follows is the host's)."""
    if not (label_u8.is_cuda and label_u8.dtype == torch.uint8 and label_u8.dim() == 2 and label_u8.is_contiguous()):
        raise RuntimeError("device_label_hists: a contiguous (H, W) uint8 label map on the device is required")
    K = len(boxes)
    H, W = label_u8.shape
    hist = torch.empty((K, 256), dtype=torch.int32, device=label_u8.device)
    arr = (np.asarray(boxes, dtype=np.int32).reshape(K, 4)).copy()
    with on_device(label_u8.device):
        rc = _lib.load_library().rfn_crop_label_hist_u8(ptr(label_u8), H, W, arr.ctypes.data, K, ptr(hist), current_stream(label_u8.device))
    _lib.check(rc, "crop_label_hist_u8")
    return hist.cpu().numpy().astype(np.int64)


def crop_flip_normalize(image_u8, label_u8, top, left, h, w, flip, out_image=None, out_label=None, mean=IMNET_MEAN, std=IMNET_STD):
    """crop + RandomHorizontalFlip + ConvertImageDtype + Normalize of a uint8 (C, H, W) image / (H, W) label map on the device,
    written into `out_image` (C, h, w) fp32 / `out_label` (h, w) int64 (slots of a batch tensor) or fresh tensors."""
    ref = image_u8 if image_u8 is not None else label_u8
    if not ref.is_cuda:
        raise RuntimeError("crop_flip_normalize: device tensors required (the product path has no CPU fallback)")
    if image_u8 is not None:
        if not (image_u8.dtype == torch.uint8 and image_u8.dim() == 3 and image_u8.is_contiguous()):
            raise RuntimeError("crop_flip_normalize: image must be a contiguous (C, H, W) uint8 tensor")
        C, H, W = image_u8.shape
        if out_image is None:
            out_image = torch.empty((C, h, w), dtype=torch.float32, device=ref.device)
        if not (out_image.dtype == torch.float32 and tuple(out_image.shape) == (C, h, w) and out_image.is_contiguous()):
            raise RuntimeError("crop_flip_normalize: out_image must be a contiguous (C, h, w) float32 tensor")
    else:
        C, (H, W) = 0, label_u8.shape
    if label_u8 is not None:
        if not (label_u8.dtype == torch.uint8 and tuple(label_u8.shape) == (H, W) and label_u8.is_contiguous()):
            raise RuntimeError("crop_flip_normalize: label must be a contiguous (H, W) uint8 tensor of the image's size")
        if out_label is None:
            out_label = torch.empty((h, w), dtype=torch.int64, device=ref.device)
        if not (out_label.dtype == torch.int64 and tuple(out_label.shape) == (h, w) and out_label.is_contiguous()):
            raise RuntimeError("crop_flip_normalize: out_label must be a contiguous (h, w) int64 tensor")
    m = np.asarray(mean, dtype=np.float32).copy()
    s = np.asarray(std, dtype=np.float32).copy()
    with on_device(ref.device):
        rc = _lib.load_library().rfn_crop_flip_norm_u8(ptr(image_u8), ptr(label_u8), C, H, W, int(top), int(left), int(h), int(w),
                                                       1 if flip else 0, m.ctypes.data, s.ctypes.data, ptr(out_image), ptr(out_label),
                                                       current_stream(ref.device))
    _lib.check(rc, "crop_flip_norm_u8")
    return out_image, out_label


def _get_params(h, w, size):
    """RandomCrop.get_params (transforms.py:340-350): no draw when the image already has the crop's size"""
    th, tw = size
    if w == tw and h == th:
        return 0, 0, h, w
    i = random.randint(0, max(h - th, 0))
    j = random.randint(0, max(w - tw, 0))
    return i, j, min(th, h), min(tw, w)


def draw_crop(h, w, size, cat_max_ratio=1.0, ignore_index=255, hists=None):
    """RandomCrop.forward's choice of the box (transforms.py:296-306) -> ((top, left, height, width), histogram of the box or None).
    cat_max_ratio < 1: the whole chain of up to 11 candidates is drawn first (remembering the state of the `random` stream after
    each), `hists(boxes)` counts them in one go, the chain is walked as the reference walks it and the stream is put back to
    where the reference would have left it."""
    first = _get_params(h, w, size)
    if not cat_max_ratio < 1.0:
        return first, None
    chain = [(first, random.getstate())]
    for _ in range(10):
        chain.append((_get_params(h, w, size), random.getstate()))
    hh = hists([c[0] for c in chain])
    pick = 10                                            # ten failed checks: the eleventh box is used unchecked
    for t in range(10):
        cnt = hh[t].copy()
        cnt[ignore_index] = 0
        present = cnt[cnt > 0]
        # `len(cnt) > 1 and cnt.max() / torch.sum(cnt).float() < ratio`: a float32 quotient compared with the python float
        if len(present) > 1 and float(np.float32(present.max()) / np.float32(present.sum())) < cat_max_ratio:
            pick = t
            break
    random.setstate(chain[pick][1])
    return chain[pick][0], hh[pick]


class RareClassSourceSampler:
    """Cityscapes.__getitem__ with rcs_enabled (datasets/cityscapes.py:100-158) over ToTensor / RandomCrop / RandomHorizontalFlip /
    ConvertImageDtype / Normalize.  `load(index)` -> (image uint8 (3, H, W), label uint8 (H, W)) host tensors (what ToTensor leaves,
    transforms.py:250-279) -- pinned ones are uploaded asynchronously.  rcs_classes / rcs_classprob / indices_with_class: as
    Cityscapes.__init__ builds them (:82-98,160-190)."""

    def __init__(self, load, rcs_classes, rcs_classprob, indices_with_class, crop_size, device, cat_max_ratio=0.75,
                 rcs_min_pixels=3000, rcs_min_crop_ratio=0.5, ignore_index=255, mean=IMNET_MEAN, std=IMNET_STD, hists=None):
        self.load, self.device = load, torch.device(device)
        self.rcs_classes, self.rcs_classprob = list(rcs_classes), rcs_classprob
        self.indices_with_class = indices_with_class
        self.size, self.cat_max_ratio, self.ignore_index = tuple(crop_size), cat_max_ratio, ignore_index
        self.rcs_min_pixels, self.rcs_min_crop_ratio = rcs_min_pixels, rcs_min_crop_ratio
        self.mean, self.std = mean, std
        self._hists = hists                               # tests inject a host counter; None: the device kernel

    def _augment_params(self, lbl_dev, h, w):
        """one `load_and_augment_sample`: the crop box (with its histogram) and the flip draw"""
        counter = self._hists if self._hists is not None else (lambda boxes: device_label_hists(lbl_dev, boxes))
        need_hist = self.cat_max_ratio < 1.0
        box, hist = draw_crop(h, w, self.size, self.cat_max_ratio, self.ignore_index, counter)
        flip = random.random() < 0.5
        if hist is None and self.rcs_min_crop_ratio > 0 and not need_hist:
            hist = counter([box])[0]
        return box, flip, hist

    def draw(self):
        """the random part of one sample -> (index, image_dev, label_dev, box, flip)"""
        c = random.choices(self.rcs_classes, weights=self.rcs_classprob, k=1)[0]
        index = random.choice(self.indices_with_class[c])
        img, lbl = self.load(index)
        img_d = img.to(self.device, non_blocking=True)
        lbl_d = lbl.to(self.device, non_blocking=True)
        self._host_label = lbl                             # (for an injected host counter)
        h, w = lbl.shape
        box, flip, hist = self._augment_params(lbl_d, h, w)
        if self.rcs_min_crop_ratio > 0:
            for _ in range(10):
                if int(hist[c]) > self.rcs_min_pixels * self.rcs_min_crop_ratio:
                    break
                box, flip, hist = self._augment_params(lbl_d, h, w)   # "a new random crop" of the same image
        return index, img_d, lbl_d, box, flip

    def sample(self, out_image=None, out_label=None):
        _, img_d, lbl_d, (top, left, hh, ww), flip = self.draw()
        return crop_flip_normalize(img_d, lbl_d, top, left, hh, ww, flip, out_image, out_label, self.mean, self.std)


class PairSampler:
    """the target set's pipeline (refign_hrda_star.yaml:25-40): RandomCrop(size) + RandomHorizontalFlip with ONE set of parameters
    for image and image_ref, then conversion + normalisation.  `load(index)` -> (image uint8, image_ref uint8) host tensors."""

    def __init__(self, load, crop_size, device, mean=IMNET_MEAN, std=IMNET_STD):
        self.load, self.size, self.device, self.mean, self.std = load, tuple(crop_size), torch.device(device), mean, std

    def sample(self, index, out_image=None, out_ref=None):
        img, ref = self.load(index)
        img_d, ref_d = img.to(self.device, non_blocking=True), ref.to(self.device, non_blocking=True)
        h, w = img.shape[-2:]
        (top, left, hh, ww), _ = draw_crop(h, w, self.size)
        flip = random.random() < 0.5
        a, _ = crop_flip_normalize(img_d, None, top, left, hh, ww, flip, out_image, None, self.mean, self.std)
        b, _ = crop_flip_normalize(ref_d, None, top, left, hh, ww, flip, out_ref, None, self.mean, self.std)
        return a, b


def merge_batches(batch):
    """CombinedDataModule.on_before_batch_transfer in training (combined_data_module.py:263-310) for sub-batches that still arrive
    as separate dicts: the supervised one carries 'semantic', the adaptation one 'image' (+ 'image_ref')."""
    src_inp, src_y, trg_inp, ref_inp = [], [], [], []
    for sub in batch:
        if 'semantic' in sub:
            src_inp.append(sub['image'])
            src_y.append(sub['semantic'])
        else:
            if 'image' in sub:
                trg_inp.append(sub['image'])
            if 'image_ref' in sub:
                ref_inp.append(sub['image_ref'])
    out = {}
    if src_inp:
        out['image_src'] = torch.cat(src_inp, dim=0)
        out['semantic_src'] = torch.cat(src_y, dim=0)
    if trg_inp:
        out['image_trg'] = torch.cat(trg_inp, dim=0)
    if ref_inp:
        out['image_ref'] = torch.cat(ref_inp, dim=0)
    return out


class UDABatchAssembler:
    """{image_src, semantic_src, image_trg, image_ref} assembled ON the device: every sample's crop is written into its slot of
    the batch tensors (no torch.cat, no host-side float images), on a side stream, into one of two buffer sets -- so that batch
    n + 1 is built while step n runs and can be handed to `Trainer.step(batch, next_batch=...)`."""

    def __init__(self, source, pairs, batch_size, device):
        self.source, self.pairs, self.b, self.device = source, pairs, int(batch_size), torch.device(device)
        th, tw = source.size
        self._sets = [{"image_src": torch.empty((self.b, 3, th, tw), dtype=torch.float32, device=self.device),
                       "semantic_src": torch.empty((self.b, th, tw), dtype=torch.int64, device=self.device),
                       "image_trg": torch.empty((self.b, 3, *pairs.size), dtype=torch.float32, device=self.device),
                       "image_ref": torch.empty((self.b, 3, *pairs.size), dtype=torch.float32, device=self.device)} for _ in range(2)]
        self._turn = 0
        self._stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._read_done = [None, None]       # per buffer set: event behind the last step that read it (consumed())

    def assemble(self, pair_indices):
        """-> (batch dict, event): the batch is complete once `event` has fired (wait on it from the consuming stream)"""
        if len(pair_indices) != self.b:
            raise RuntimeError("UDABatchAssembler: one target index per sample of the batch")
        out = self._sets[self._turn]
        done, self._read_done[self._turn] = self._read_done[self._turn], None
        self._turn ^= 1
        # the set being re-filled was handed out two batches ago: the crop kernels must not overwrite it while the step that
        # read it may still be running.  If the consumer marked that step (consumed()), wait for exactly it -- batch
        # n + 1 is then built NEXT TO step n; otherwise for everything queued on the consumer's stream so far
        if done is not None:
            self._stream.wait_event(done)
        else:
            self._stream.wait_stream(torch.cuda.current_stream(self.device))
        ctx = torch.cuda.stream(self._stream)
        with ctx:
            for i in range(self.b):                          # the source loader's samples, then the target loader's (two loaders)
                self.source.sample(out["image_src"][i], out["semantic_src"][i])
            for i, idx in enumerate(pair_indices):
                self.pairs.sample(idx, out["image_trg"][i], out["image_ref"][i])
            ev = self._stream.record_event()
        return out, ev

    def consumed(self, batch):
        """Call on the consuming stream right after queueing the last work that reads `batch` (a dict assemble() returned): the
        set may be re-filled as soon as that work is done."""
        for i, st in enumerate(self._sets):
            if batch is st or batch.get("image_src") is st["image_src"]:
                self._read_done[i] = torch.cuda.current_stream(self.device).record_event()
                return
        raise RuntimeError("UDABatchAssembler.consumed: not a batch of this assembler")
