"""Halo-tiled 3 x 3 convolution (refign_amd/csrc/conv3x3.hip, behind rfn_conv2d_nhwc): against torch fp32 on the operands' 16-bit
values -- ragged sizes (tiles overhanging the image on both axes), both tile widths (N % 128 == 0 / N % 64 == 0), one and several
64-channel chunks, every activation, bf16 and f16 -- and repeatable under memory load (counted LDS-DMA hand-offs)."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


CASES = [  # B, H, W, C, N, dtype, act
    (2, 37, 61, 64, 64, torch.float16, 'relu'),
    (1, 16, 32, 128, 128, torch.float16, 'relu'),
    (2, 135, 240, 256, 256, torch.bfloat16, None),
    (3, 40, 70, 128, 64, torch.bfloat16, 'leaky'),
    (1, 9, 200, 64, 192, torch.float16, None),
    (1, 33, 33, 192, 128, torch.bfloat16, 'relu'),
    (2, 8, 512, 64, 128, torch.float16, None),
]


def _ref(x, w, bias, dt, act):
    ref = F.conv2d(x.float(), w.to(dt).float(), None if bias is None else bias.to(dt).float(), padding=1)
    return F.relu(ref) if act == 'relu' else (F.leaky_relu(ref, 0.1) if act == 'leaky' else ref)


@pytest.mark.parametrize("B,H,W,C,N,dt,act", CASES)
def test_halo_tiled_conv3x3_matches_fp32(dev, B, H, W, C, N, dt, act):
    from refign_amd import conv
    torch.manual_seed(H * W + C)
    x = torch.randn(B, C, H, W, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(N, C, 3, 3, device=dev) * (9 * C) ** -0.5
    bias = torch.randn(N, device=dev) * 0.1 if act != 'leaky' else None
    with torch.no_grad():
        y = conv.conv2d_mfma(x, w, bias, 1, 1, 1, act=act, dtype=dt)
    assert y is not None and tuple(y.shape) == (B, N, H, W)
    ref = _ref(x, w, bias, dt, act)
    err = float((y.float() - ref).abs().max()) / float(ref.abs().max())
    assert err < (1.2e-2 if dt == torch.bfloat16 else 2e-3), err


def test_halo_tiled_conv3x3_is_the_kernel_that_runs_and_agrees_with_the_implicit_gemm(dev, tmp_path):
    """The same call in two fresh processes, with and without RFN_CONV_HALO=0 (the library reads the switch once): the results agree
    to the rounding of the result type, and the default one comes from conv3x3_halo_kernel (checked by its launch through the
    profiler-free route: the two results are NOT bit-identical -- another summation order)."""
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "from refign_amd import conv\n"
        "torch.manual_seed(5)\n"
        "x = torch.randn(2, 128, 48, 96, device='cuda').half().contiguous(memory_format=torch.channels_last)\n"
        "w = torch.randn(128, 128, 3, 3, device='cuda') * 0.03\n"
        "b = torch.randn(128, device='cuda') * 0.1\n"
        "with torch.no_grad():\n"
        "    y = conv.conv2d_mfma(x, w, b, 1, 1, 1, act='relu', dtype=torch.float16)\n"
        "torch.save(y.float().cpu(), sys.argv[1])\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for flag in ("1", "0"):
        path = str(tmp_path / f"y{flag}.pt")
        env = dict(os.environ, RFN_CONV_HALO=flag)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env, timeout=300)
        outs.append(torch.load(path))
    a, b = outs
    assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max())
    assert not torch.equal(a, b)


def test_halo_tiled_conv3x3_is_repeatable_under_memory_load(dev):
    from refign_amd import conv
    torch.manual_seed(11)
    x = torch.randn(4, 64, 270, 480, device=dev).half().contiguous(memory_format=torch.channels_last)
    w = torch.randn(128, 64, 3, 3, device=dev) * 0.04
    noise = torch.empty(64 << 20, device=dev)
    side = torch.cuda.Stream()
    with torch.no_grad():
        first = conv.conv2d_mfma(x, w, None, 1, 1, 1, act='relu', dtype=torch.float16).clone()
        for _ in range(8):
            with torch.cuda.stream(side):
                noise.normal_()
            y = conv.conv2d_mfma(x, w, None, 1, 1, 1, act='relu', dtype=torch.float16)
            assert torch.equal(y, first)
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,H,W,C,N,act", [(2, 50, 70, 128, 128, 3), (1, 67, 90, 96, 64, 3), (2, 40, 64, 64, 32, 0), (1, 48, 100, 128, 96, 1),
                                           (2, 135, 240, 32, 128, 3)])
def test_halo_tiled_conv3x3_fp32_result_of_split_products(dev, B, H, W, C, N, act):
    """The split-bf16 convolutions of the matcher's head (split32.conv2d_parts: 3 C channels per pixel, fp32 result and bias) on the
    halo-tiled kernel: a last half chunk (3 C % 64 == 32), output channels that do not fill the last 64-channel tile, every
    activation -- against torch fp32 (three bf16 products leave ~2^-16 relative)."""
    from refign_amd import split32
    torch.manual_seed(C + N)
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(N, C, 3, 3, device=dev) * (9 * C) ** -0.5
    bias = torch.randn(N, device=dev) * 0.1
    with torch.no_grad():
        # (up to 96 channels the decoders' first layers hand over the PARTS of a concatenation: split32.conv2d_parts)
        y = split32.conv2d_parts([x[:, :C // 2], x[:, C // 2:]], w, bias, 1, 1, 1, act) if C <= 96 else \
            split32.conv2d(x, w, bias, 1, 1, 1, act)
        ref = F.conv2d(x, w, bias, padding=1)
    assert y is not None and tuple(y.shape) == (B, N, H, W)
    ref = F.relu(ref) if act == 1 else (F.leaky_relu(ref, 0.1) if act == 3 else ref)
    err = float((y - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-4, err
