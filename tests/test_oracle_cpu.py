"""CPU: the oracle (oracle/) against the golden vectors captured from the reference, and -- when the compiled reference
oracle/_ref is present -- against the reference C++ itself on fresh random inputs."""
import numpy as np
import pytest
from conftest import golden, golden_names


@pytest.mark.parametrize("name", golden_names("corr_"))
def test_corr_oracle_matches_reference_golden(oracle, name):
    g = golden(name)
    a = [int(v) for v in g["args"]]
    kw = dict(kernel_size=(a[0], a[1]), patch_size=(a[2], a[3]), padding=(a[4], a[5]), dilation=(a[6], a[7]),
              dilation_patch=(a[8], a[9]), stride=(a[10], a[11]))
    out = oracle.corr_forward(g["in1"], g["in2"], **kw)
    # same loop nest, same accumulation order, no FMA contraction => bit-exact
    np.testing.assert_array_equal(out, g["out"])
    g1, g2 = oracle.corr_backward(g["in1"], g["in2"], g["grad_out"], **kw)
    np.testing.assert_array_equal(g1, g["grad_in1"])
    np.testing.assert_array_equal(g2, g["grad_in2"])


def test_corr_oracle_matches_compiled_reference(oracle):
    import build_ref
    ref = build_ref.load_prebuilt()
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    import torch
    rng = np.random.default_rng(7)
    for (B, C, H, W, args) in [(2, 7, 19, 23, (1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)),
                               (1, 3, 14, 15, (3, 3, 7, 5, 2, 1, 1, 2, 2, 1, 2, 1))]:
        a = rng.standard_normal((B, C, H, W)).astype(np.float32)
        b = rng.standard_normal((B, C, H, W)).astype(np.float32)
        want = ref.forward(torch.from_numpy(a), torch.from_numpy(b), *args).numpy()
        kw = dict(kernel_size=args[0:2], patch_size=args[2:4], padding=args[4:6], dilation=args[6:8],
                  dilation_patch=args[8:10], stride=args[10:12])
        np.testing.assert_array_equal(oracle.corr_forward(a, b, **kw), want)
        go = rng.standard_normal(want.shape).astype(np.float32)
        w1, w2 = ref.backward(torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(go), *args)
        g1, g2 = oracle.corr_backward(a, b, go, **kw)
        np.testing.assert_array_equal(g1, w1.numpy())
        np.testing.assert_array_equal(g2, w2.numpy())


@pytest.mark.parametrize("name", golden_names("localcorr_"))
def test_local_layer_oracle(oracle, name):
    g = golden(name)
    out = oracle.local_correlation_layer(g["source"], g["target"])
    np.testing.assert_allclose(out, g["out"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["globalcorr_c64_16x16", "globalcorr_c24_5x7_6x4"])
def test_global_layer_oracle(oracle, name):
    g = golden(name)
    out = oracle.global_correlation_layer(g["source"], g["target"])
    np.testing.assert_allclose(out, g["out"], rtol=2e-4, atol=2e-6)


def test_global_layer_oracle_level4(oracle):
    from fill import hashed_uniform
    g = golden("globalcorr_c512_level4")
    src = oracle.l2_normalize(hashed_uniform((2, 512, 16, 16), "g3/src") - 0.5)
    trg = oracle.l2_normalize(hashed_uniform((2, 512, 16, 16), "g3/trg") - 0.5)
    out = oracle.global_correlation_layer(src, trg)
    np.testing.assert_allclose(out[:, ::7, ::3, ::5], g["out_sample"], rtol=5e-4, atol=5e-6)
    assert abs(out.astype(np.float64).sum() - g["checksum"]) < 1e-3 * g["abs_checksum"]


@pytest.mark.parametrize("name", golden_names("warp_"))
def test_warp_oracle(oracle, name):
    g = golden(name)
    out, mask = oracle.warp(g["x"], g["flow"], return_mask=True)
    np.testing.assert_allclose(out, g["out"], rtol=1e-5, atol=2e-5)
    np.testing.assert_array_equal(mask, g["mask"])


def test_matching_misc_oracle(oracle):
    g = golden("matching_misc")
    np.testing.assert_allclose(oracle.unnormalise_and_convert_mapping_to_flow(g["mapping"]), g["flow"], atol=1e-5)
    np.testing.assert_allclose(oracle.confidence_from_logvar(g["logvar"]), g["conf"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", golden_names("refine_"))
def test_refine_oracle(oracle, name):
    g = golden(name)
    if bool(g["no_mask"]):
        out = oracle.refine(g["logits_trg"], g["logits_ref"], None, None, gamma=float(g["gamma"]))
    else:
        out = oracle.refine(g["logits_trg"], g["logits_ref"], g["mask"], g["cert"], gamma=float(g["gamma"]),
                            disable_M=bool(g["disable_M"]), disable_P=bool(g["disable_P"]))
    np.testing.assert_allclose(out, g["out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(oracle.eta(g["logits_trg"]), g["eta"], rtol=1e-5, atol=1e-6)
    prob, label, weight = oracle.pseudo_label(g["out"])
    np.testing.assert_array_equal(label, g["pseudo_label"])
    np.testing.assert_array_equal(prob, g["pseudo_prob"])
    assert abs(weight - g["pseudo_weight"]) < 1e-7


# ------------------------------------------------------------------ head / align restatement (oracle/cpu_align.py)
def _unit(shape, key):
    from fill import hashed_uniform
    x = hashed_uniform(shape, key) - 0.5
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def test_cpu_head_restatement_matches_reference_golden(oracle):
    """The CPU restatement of the UAWarpC head (test infrastructure used by bench.py's cpu_baseline) reproduces the
    reference's four (flow, log-variance) levels on the G5 fixture."""
    import torch
    import cpu_align
    from fill import closed_form_fill
    from refign_amd.align import UAWarpCHead
    name, H, W = "rect_192x320", 192, 320
    g = golden("head_" + name)
    head = closed_form_fill(UAWarpCHead(in_index=[0, 1], input_transform='multiple_select',
                                        estimate_uncertainty=True)).eval()
    p = {"trg": [_unit((1, 128, H // 4, W // 4), f"g5/{name}/t1"), _unit((1, 256, H // 8, W // 8), f"g5/{name}/t2")],
         "src": [_unit((1, 128, H // 4, W // 4), f"g5/{name}/s1"), _unit((1, 256, H // 8, W // 8), f"g5/{name}/s2")],
         "trg256": [_unit((1, 256, 32, 32), f"g5/{name}/t3"), _unit((1, 512, 16, 16), f"g5/{name}/t4")],
         "src256": [_unit((1, 256, 32, 32), f"g5/{name}/s3"), _unit((1, 512, 16, 16), f"g5/{name}/s4")]}
    for k_t, k_s in (("trg", "src"), ("trg256", "src256")):
        for i in range(2):
            mix = 0.7 * np.roll(p[k_t][i], shift=(1, -2), axis=(2, 3)) + 0.3 * p[k_s][i]
            p[k_s][i] = (mix / np.linalg.norm(mix, axis=1, keepdims=True)).astype(np.float32)
    t = lambda xs: [torch.from_numpy(x) for x in xs]  # noqa: E731
    corr_fn = lambda a, b: torch.from_numpy(oracle.corr_forward(a.contiguous().numpy(), b.contiguous().numpy(), patch_size=9))  # noqa: E731
    outs = cpu_align.head_forward(head, t(p["trg"]), t(p["src"]), t(p["trg256"]), t(p["src256"]), (H, W), corr_fn)
    for lvl, (fl, un) in zip((4, 3, 2, 1), outs):
        np.testing.assert_allclose(fl.numpy(), g[f"flow{lvl}"], rtol=1e-3, atol=2e-2, err_msg=f"flow{lvl}")
        np.testing.assert_allclose(un.numpy(), g[f"uncert{lvl}"], rtol=1e-3, atol=5e-3, err_msg=f"uncert{lvl}")


def test_cpu_alignment_forward_restatement_matches_reference_k2_golden():
    """oracle/cpu_align.alignment_forward (AlignmentModel.forward, models/alignment_model.py:55-79, restated on torch-CPU ops +
    the correlation oracle) against the reference's own output on the K2 input (tests/golden/make_golden_k4.py K2): the CPU
    baseline of `bench.py --workload uawarpc_align_512x512` is a pinned computation."""
    import os
    import sys
    import numpy as np
    import torch
    from conftest import golden
    from fill import closed_form_fill, hashed_uniform
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import cpu_align
    from refign_amd.align import VGG, UAWarpCHead
    g = golden("alignment_forward_512x512")
    B, H, W = [int(v) for v in g["size"]]
    vgg = closed_form_fill(VGG('vgg16', out_indices=[2, 3, 4]), "alignment_backbone.").eval()
    head = closed_form_fill(UAWarpCHead(in_index=[0, 1], input_transform='multiple_select', estimate_uncertainty=True)).eval()
    img_i = (hashed_uniform((B, 3, H, W), "k2/i") * 4 - 2).astype(np.float32)
    img_j = (0.8 * np.roll(img_i, (3, -2), (2, 3)) + 0.2 * (hashed_uniform((B, 3, H, W), "k2/j") * 4 - 2)).astype(np.float32)
    flow, uncert = cpu_align.alignment_forward(vgg, head, torch.from_numpy(img_i), torch.from_numpy(img_j))
    np.testing.assert_allclose(flow[:, :, ::4, ::4].numpy(), g["flow_sample"], atol=2e-3)
    np.testing.assert_allclose(uncert[:, :, ::4, ::4].numpy(), g["uncert_sample"], atol=2e-4)
