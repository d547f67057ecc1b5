"""tests/golden/make_golden.py -- regenerate the golden vectors (authoring container only).

Imports the REFERENCE (brdav/refign, /root/reference) through tests/golden/_ref_import.py, runs its own functions on
seeded inputs and stores inputs + outputs as small .npz fixtures next to this script.  The fixtures are DATA (tensors);
no reference source is stored.  Tests never import the reference: they read these files.

    python tests/golden/make_golden.py            # all groups
    python tests/golden/make_golden.py G1 G8      # selected groups

Groups:  G1 correlation fwd/bwd (compiled reference C++), G2 LocalFeatureCorrelationLayer,
G3 GlobalFeatureCorrelationLayer, G6 warp (+mask, out-of-range, zero flow), G8 refine + pseudo-label statistics,
G4 decoder / refinement / uncertainty modules, G5 UAWarpCHead, G7 align() end to end (closed-form weights,
tests/golden/fill.py).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as R  # noqa: E402
from fill import closed_form_fill, hashed_uniform  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} kB)")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def g1(corr):
    """Reference C++ sampler (correlation.cpp) forward + backward."""
    rng = np.random.default_rng(101)
    cases = [
        # name, B, C, H, W, (kH,kW), (pH,pW), (padH,padW), (dilH,dilW), (dpH,dpW), (sH,sW)
        ("hot_c3_5x7", 2, 3, 5, 7, (1, 1), (9, 9), (0, 0), (1, 1), (1, 1), (1, 1)),
        ("hot_c16_32x32", 1, 16, 32, 32, (1, 1), (9, 9), (0, 0), (1, 1), (1, 1), (1, 1)),
        ("hot_c19_21x37", 1, 19, 21, 37, (1, 1), (9, 9), (0, 0), (1, 1), (1, 1), (1, 1)),
        ("hot_c40_9x68", 1, 40, 9, 68, (1, 1), (9, 9), (0, 0), (1, 1), (1, 1), (1, 1)),
        ("gen_k3_p5_s2", 2, 4, 11, 13, (3, 3), (5, 5), (1, 1), (1, 1), (2, 2), (2, 2)),
        ("gen_k1_p3_dp2", 1, 5, 9, 10, (1, 1), (3, 3), (0, 0), (1, 1), (2, 2), (1, 1)),
        ("gen_k3_dil2_pad2_rect", 1, 3, 12, 9, (3, 1), (3, 5), (2, 0), (2, 1), (1, 2), (1, 2)),
    ]
    for name, B, C, H, W, k, p, pad, dil, dp, s in cases:
        a = rng.standard_normal((B, C, H, W)).astype(np.float32)
        b = rng.standard_normal((B, C, H, W)).astype(np.float32)
        args = (k[0], k[1], p[0], p[1], pad[0], pad[1], dil[0], dil[1], dp[0], dp[1], s[0], s[1])
        out = corr.forward(t(a), t(b), *args)
        go = rng.standard_normal(tuple(out.shape)).astype(np.float32)
        g1_, g2_ = corr.backward(t(a), t(b), t(go), *args)
        save("corr_" + name, in1=a, in2=b, args=np.array(args, dtype=np.int32), out=out.numpy(), grad_out=go,
             grad_in1=g1_.numpy(), grad_in2=g2_.numpy())
    # float64 case (CPU reference dispatches double too)
    a = rng.standard_normal((1, 6, 10, 11))
    b = rng.standard_normal((1, 6, 10, 11))
    args = (1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)
    out = corr.forward(t(a), t(b), *args)
    go = rng.standard_normal(tuple(out.shape))
    g1_, g2_ = corr.backward(t(a), t(b), t(go), *args)
    save("corr_hot_f64", in1=a, in2=b, args=np.array(args, dtype=np.int32), out=out.numpy(), grad_out=go,
         grad_in1=g1_.numpy(), grad_in2=g2_.numpy())


def g2():
    mods = R.ref_module("models.modules")
    rng = np.random.default_rng(202)
    layer = mods.LocalFeatureCorrelationLayer(patch_size=9)
    for name, B, C, H, W in [("c8_12x20", 2, 8, 12, 20), ("c32_33x70", 1, 32, 33, 70)]:
        src = rng.standard_normal((B, C, H, W)).astype(np.float32)
        trg = rng.standard_normal((B, C, H, W)).astype(np.float32)
        src /= np.linalg.norm(src, axis=1, keepdims=True)
        trg /= np.linalg.norm(trg, axis=1, keepdims=True)
        out = layer(t(src), t(trg))
        save("localcorr_" + name, source=src, target=trg, out=out.numpy())


def g3():
    mods = R.ref_module("models.modules")
    rng = np.random.default_rng(303)
    layer = mods.GlobalFeatureCorrelationLayer(cyclic_consistency=True)
    for name, B, C, hs, ws, ht, wt in [("c64_16x16", 2, 64, 16, 16, 16, 16), ("c24_5x7_6x4", 1, 24, 5, 7, 6, 4)]:
        src = rng.standard_normal((B, C, hs, ws)).astype(np.float32)
        trg = rng.standard_normal((B, C, ht, wt)).astype(np.float32)
        src /= np.linalg.norm(src, axis=1, keepdims=True)
        trg /= np.linalg.norm(trg, axis=1, keepdims=True)
        out = layer(t(src), t(trg))
        save("globalcorr_" + name, source=src, target=trg, out=out.numpy())
    # level-4 sized case (C=512, 16x16): store inputs by generator, output as strided sample + checksum
    src = hashed_uniform((2, 512, 16, 16), "g3/src") - 0.5
    trg = hashed_uniform((2, 512, 16, 16), "g3/trg") - 0.5
    out = layer(torch.nn.functional.normalize(t(src), dim=1), torch.nn.functional.normalize(t(trg), dim=1)).numpy()
    save("globalcorr_c512_level4", out_sample=out[:, ::7, ::3, ::5], checksum=np.float64(out.astype(np.float64).sum()),
         abs_checksum=np.float64(np.abs(out.astype(np.float64)).sum()))


def g6():
    mu = R.ref_module("helpers.matching_utils")
    rng = np.random.default_rng(606)
    for name, B, C, H, W, mag in [("c5_9x11", 2, 5, 9, 11, 3.0), ("c19_24x40_big", 1, 19, 24, 40, 25.0)]:
        x = rng.standard_normal((B, C, H, W)).astype(np.float32)
        flo = (mag * rng.standard_normal((B, 2, H, W))).astype(np.float32)
        flo[:, :, 0, 0] = 0.0            # exactly-on-grid sample
        flo[0, 0, 1, 1] = 1e6            # far outside
        flo[0, 1, 2, 2] = -1e6
        out, mask = mu.warp(t(x), t(flo), return_mask=True)
        save("warp_" + name, x=x, flow=flo, out=out.numpy(), mask=mask.numpy())
    x = rng.standard_normal((1, 3, 6, 8)).astype(np.float32)
    flo = np.zeros((1, 2, 6, 8), dtype=np.float32)
    out, mask = mu.warp(t(x), t(flo), return_mask=True)
    save("warp_zero_flow", x=x, flow=flo, out=out.numpy(), mask=mask.numpy())
    # mapping -> flow and confidence
    m = rng.uniform(-1.2, 1.2, (2, 2, 16, 16)).astype(np.float32)
    lv = rng.uniform(-6, 6, (2, 1, 8, 9)).astype(np.float32)
    save("matching_misc", mapping=m, flow=mu.unnormalise_and_convert_mapping_to_flow(t(m)).numpy(), logvar=lv,
         conf=mu.estimate_probability_of_confidence_interval_of_mixture_density(t(lv)).numpy())


def g8():
    sm = R.ref_module("models.segmentation_model")
    Model = sm.DomainAdaptationSegmentationModel
    rng = np.random.default_rng(808)
    for name, B, H, W, flags in [("b2_16x24", 2, 16, 24, {}), ("b1_9x13_noM", 1, 9, 13, {"disable_M": True}),
                                 ("b1_9x13_noP", 1, 9, 13, {"disable_P": True}),
                                 ("b2_8x8_nomask", 2, 8, 8, {"no_mask": True})]:
        lt = (3.0 * rng.standard_normal((B, 19, H, W))).astype(np.float32)
        lr = (3.0 * rng.standard_normal((B, 19, H, W))).astype(np.float32)
        # make a good share of pixels agree on a static class so that M is exercised
        agree = rng.random((B, H, W)) < 0.5
        cls = rng.choice([0, 1, 2, 3, 4, 8, 9, 10, 5, 13], size=(B, H, W))
        for c in range(19):
            lt[:, c][agree & (cls == c)] += 8.0
            lr[:, c][agree & (cls == c)] += 8.0
        mask = rng.random((B, H, W)) < 0.8
        cert = rng.random((B, 1, H, W)).astype(np.float32)
        ns = types.SimpleNamespace(gamma=0.25, disable_M=flags.get("disable_M", False),
                                   disable_P=flags.get("disable_P", False), eta=Model.eta)
        if flags.get("no_mask"):
            out = Model.refine(ns, t(lt), t(lr), None, None)
        else:
            out = Model.refine(ns, t(lt), t(lr), t(mask), t(cert))
        # pseudo-label statistics of get_dacs_mix (segmentation_model.py:551-556)
        prob, label = torch.max(out, dim=1)
        weight = torch.sum(prob.ge(0.968).long() == 1) / torch.numel(label)
        save("refine_" + name, logits_trg=lt, logits_ref=lr, mask=mask, cert=cert, out=out.numpy(),
             eta=Model.eta(t(lt)).numpy(), pseudo_prob=prob.numpy(), pseudo_label=label.numpy(),
             pseudo_weight=np.float32(weight.item()), gamma=np.float32(0.25),
             disable_M=np.bool_(ns.disable_M), disable_P=np.bool_(ns.disable_P),
             no_mask=np.bool_(flags.get("no_mask", False)))


GROUPS = {"G1": g1, "G2": g2, "G3": g3, "G6": g6, "G8": g8}


def main(argv):
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    corr = R.setup()
    import make_golden_modules as mm   # G4, G5, G7 (need the closed-form weight filler)
    import make_golden_seg as ms       # G9-G12
    import make_golden_step as mst     # G13
    GROUPS.update(mm.GROUPS)
    GROUPS.update(ms.GROUPS)
    GROUPS.update(mst.GROUPS)
    which = argv or sorted(GROUPS)
    for g in which:
        print(g)
        fn = GROUPS[g]
        fn(corr) if g == "G1" else fn()


if __name__ == "__main__":
    main(sys.argv[1:])
