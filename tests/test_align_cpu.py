"""CPU: the align-path modules expose exactly the reference's state_dict keys and shapes (so reference checkpoints load
with strict=True) and the VGG tap indices match."""
import json
import os

from conftest import GOLDEN


def _manifest():
    return json.load(open(os.path.join(GOLDEN, "state_dict_manifest.json")))


def test_uawarpc_head_state_dict_matches_reference():
    from refign_amd.align import UAWarpCHead
    head = UAWarpCHead(in_index=[0, 1], input_transform='multiple_select', estimate_uncertainty=True)
    want = _manifest()["UAWarpCHead(estimate_uncertainty=True)"]
    got = {k: list(v.shape) for k, v in head.state_dict().items()}
    assert got == want


def test_vgg_state_dict_and_taps_match_reference():
    from refign_amd.align import VGG
    vgg = VGG('vgg16', out_indices=[2, 3, 4])
    want = _manifest()["VGG(vgg16)"]
    assert {k: list(v.shape) for k, v in vgg.state_dict().items()} == want
    assert vgg.layer_indices == [10, 17, 24]


def test_closed_form_fill_is_deterministic():
    from fill import closed_form_fill, hashed_uniform
    from refign_amd.align import RefinementModule
    a = closed_form_fill(RefinementModule(32), "x.").state_dict()
    b = closed_form_fill(RefinementModule(32), "x.").state_dict()
    assert all((a[k] == b[k]).all() for k in a)
    u = hashed_uniform((1000,), "k")
    assert 0.45 < float(u.mean()) < 0.55 and float(u.min()) >= 0 and float(u.max()) < 1


def test_uncertainty_micro_image_chain_as_tiled_convolutions():
    import torch
    """UncertaintyModule._patch_statistics_tiled (the autograd path of matcher training: four convolutions on tiled
    images) == the reference's formulation (one convolution chain over B*H*W 9x9 micro-images, modules.py:529-551) in
    train mode: outputs, gradients with respect to the correlation volume and every parameter, BatchNorm buffers."""
    import copy
    from refign_amd.align import UncertaintyModule
    torch.manual_seed(0)
    m = UncertaintyModule(1, search_size=9, feed_in_previous=True).double().train()
    m2 = copy.deepcopy(m)
    corr = torch.rand(2, 81, 5, 6, dtype=torch.float64).requires_grad_(True)
    a = m._patch_statistics_tiled(corr)
    b, _, h, w = corr.shape
    x = corr.permute(0, 2, 3, 1).reshape(b * h * w, 1, 9, 9)
    ref = m2.predict_uncertainty(m2.conv_2(m2.conv_1(m2.conv_0(x)))).flatten(1).view(b, h, w, 6).permute(0, 3, 1, 2)
    assert torch.allclose(a, ref, atol=1e-12)
    ga = torch.autograd.grad(a.square().sum(), [corr] + list(m.parameters()), allow_unused=True)
    gb = torch.autograd.grad(ref.square().sum(), [corr] + list(m2.parameters()), allow_unused=True)
    n = 0
    for x_, y_ in zip(ga, gb):
        assert (x_ is None) == (y_ is None)
        if x_ is not None:
            assert torch.allclose(x_, y_, atol=1e-10)
            n += 1
    assert n >= 10
    for k, v in m.state_dict().items():
        assert torch.allclose(v.double(), m2.state_dict()[k].double(), atol=1e-12), k
