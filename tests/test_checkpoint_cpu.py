"""N3 -- checkpoint interchange (models/segmentation_model.py:421-436, models/heads/uawarpc.py:282-305,
models/backbones/mix_transformer.py:445-479): Lightning-format `.ckpt` files ({'state_dict': ...}) written by the
reference load into refign_amd with strict=True and the other way round, with identical tensors; the sub-module loaders
strip the prefixes the reference strips (`alignment_head.`, `backbone.`, drop `head.*`)."""
import os
import sys

import pytest
import torch

HAVE_REF = os.path.isdir("/root/reference/models")
DIMS = [32, 64, 160, 256]
OPT = {"class_path": "torch.optim.AdamW", "init_args": {"lr": 6e-5, "weight_decay": 0.01}}
SCH = {"class_path": "helpers.lr_scheduler.LinearWarmupPolynomialLR",
       "init_args": {"warmup_iters": 1500, "warmup_ratio": 1e-6, "power": 1.0, "max_steps": 40000}}


def ours(use_hrda=True, pretrained=None):
    from refign_amd.align import VGG, UAWarpCHead
    from refign_amd.seg import DAFormerHead, MixVisionTransformer, PixelWeightedCrossEntropyLoss, SegFormerHead
    from refign_amd.uda import DomainAdaptationSegmentationModel
    return DomainAdaptationSegmentationModel(
        OPT, SCH, backbone=MixVisionTransformer("mit_b0"), head=DAFormerHead(DIMS, [0, 1, 2, 3], 19, 'multiple_select'),
        loss=PixelWeightedCrossEntropyLoss(), alignment_backbone=VGG('vgg16', out_indices=[2, 3, 4]),
        alignment_head=UAWarpCHead(in_index=[0, 1], input_transform='multiple_select', estimate_uncertainty=True),
        use_refign=True, use_hrda=use_hrda, hrda_scale_attention=SegFormerHead(DIMS, [0, 1, 2, 3], 19, 'multiple_select'),
        pretrained=pretrained)


def test_lightning_checkpoint_round_trip(tmp_path):
    """save -> load(strict) -> identical state, through the `pretrained=` constructor keyword of the reference; the
    UAWarpC head and the MiT backbone pick their sub-trees out of the same file."""
    from fill import closed_form_fill
    from refign_amd.align import UAWarpCHead
    from refign_amd.seg import MixVisionTransformer
    a = closed_form_fill(ours())
    path = str(tmp_path / "model.ckpt")
    torch.save({"state_dict": a.state_dict(), "global_step": 7, "pytorch-lightning_version": "1.5.10"}, path)
    b = ours(pretrained=path)
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb)
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    head = UAWarpCHead(in_index=[0, 1], input_transform='multiple_select', estimate_uncertainty=True, pretrained=path)
    assert all(torch.equal(v, sa["alignment_head." + k]) for k, v in head.state_dict().items())
    mit = MixVisionTransformer("mit_b0", pretrained=path)
    assert all(torch.equal(v, sa["backbone." + k]) for k, v in mit.state_dict().items())
    # a key that does not belong makes the strict load fail, like the reference
    sd = dict(sa)
    sd["backbone.bogus"] = torch.zeros(1)
    torch.save({"state_dict": sd}, path)
    with pytest.raises(RuntimeError, match="bogus"):
        ours(pretrained=path)


@pytest.mark.skipif(not HAVE_REF, reason="reference checkout not present (authoring container only)")
def test_checkpoints_interchange_with_the_reference(tmp_path):
    """A checkpoint written from the REFERENCE's module tree loads here with strict=True, one written here loads into
    the reference with strict=True, tensor for tensor."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden_step as G          # builds the reference model through tests/golden/_ref_import.py
    import _ref_import as R
    from fill import closed_form_fill
    R.setup()
    ref = G.run_reference(True)           # HRDA variant, closed-form weights
    path = str(tmp_path / "ref.ckpt")
    torch.save({"state_dict": ref.state_dict()}, path)
    mine = ours(pretrained=path)
    sr, sm = ref.state_dict(), mine.state_dict()
    assert set(sr) == set(sm), sorted(set(sr) ^ set(sm))[:10]
    assert all(sr[k].shape == sm[k].shape and torch.equal(sr[k], sm[k]) for k in sr)
    # and back: perturb ours, save, load into the reference with its own loader
    with torch.no_grad():
        for p in mine.parameters():
            p.mul_(1.5)
    back = str(tmp_path / "ours.ckpt")
    torch.save({"state_dict": mine.state_dict()}, back)
    ref.load_weights(back)
    sr = ref.state_dict()
    assert all(torch.equal(sr[k], v) for k, v in mine.state_dict().items())
