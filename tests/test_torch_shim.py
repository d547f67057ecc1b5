"""The pybind11 / torch-extension module `correlation` (refign_amd/torch_shim): the reference's native operator boundary
(models/correlation_ops/correlation_sampler.cpp:129-132: forward / backward with 12 int arguments) as a built artefact on top
of the C ABI.  CPU: it builds, loads, exports both functions and rejects what the reference's checks reject.  GPU: called the way
models/correlation_ops/correlation_function.py:70-74,89-93 calls it, against the reference goldens (G1) and the ctypes path."""
import numpy as np
import pytest
import torch
from conftest import golden


def _mod():
    from refign_amd.torch_shim.build import load
    return load()


def test_shim_builds_loads_and_checks_inputs():
    m = _mod()
    assert callable(m.forward) and callable(m.backward)
    a = torch.randn(1, 3, 5, 7)
    with pytest.raises(RuntimeError):                          # the CPU path belongs to the reference's own correlation.cpp
        m.forward(a, a, 1, 1, 3, 3, 0, 0, 1, 1, 1, 1, 1, 1)
    with pytest.raises(TypeError):                             # 12 ints, like the reference's binding
        m.forward(a, a, 1, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["corr_hot_c3_5x7", "corr_hot_c16_32x32", "corr_gen_k3_dil2_pad2_rect"])
def test_shim_matches_reference_goldens(dev, name):
    m = _mod()
    g = golden(name)
    in1, in2 = torch.from_numpy(g["in1"]).to(dev), torch.from_numpy(g["in2"]).to(dev)
    a = [int(v) for v in g["args"]]          # kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW
    out = m.forward(in1, in2, *a)
    assert tuple(out.shape) == g["out"].shape
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    g1, g2 = m.backward(in1, in2, torch.from_numpy(g["grad_out"]).to(dev), *a)
    np.testing.assert_allclose(g1.cpu().numpy(), g["grad_in1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g2.cpu().numpy(), g["grad_in2"], rtol=1e-4, atol=1e-5)
    with pytest.raises(RuntimeError):
        m.forward(in1.transpose(2, 3), in2.transpose(2, 3), *a)   # CHECK_CONTIGUOUS
    # half, as the reference's device dispatch takes it (correlation_cuda_kernel.cu:267): fp32 sums, one rounding
    oh = m.forward(in1.half(), in2.half(), *a)
    assert oh.dtype == torch.float16
    want = m.forward(in1.half().float(), in2.half().float(), *a)
    assert float((oh.float() - want).abs().max()) <= 2.0 ** -10 * max(float(want.abs().max()), 1e-3)
    h1, h2 = m.backward(in1.half(), in2.half(), torch.from_numpy(g["grad_out"]).to(dev).half(), *a)
    w1, w2 = m.backward(in1.half().float(), in2.half().float(), torch.from_numpy(g["grad_out"]).to(dev).half().float(), *a)
    for got, ref in ((h1, w1), (h2, w2)):
        assert got.dtype == torch.float16
        assert float((got.float() - ref).abs().max()) <= 2.0 ** -10 * max(float(ref.abs().max()), 1e-3)
