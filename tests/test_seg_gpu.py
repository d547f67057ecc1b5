"""GPU run of the segmentation-network parity checks of test_seg_cpu.py (same goldens, device = cuda:0)."""
import pytest
import test_seg_cpu as S

pytestmark = pytest.mark.gpu


def test_mit_golden_gpu(dev):
    S.check_mit(dev, 1e-3)


def test_heads_golden_gpu(dev):
    S.check_heads(dev, 1e-3)


def test_hrda_golden_gpu(dev):
    S.check_hrda(dev, 2e-3)


def test_loss_golden_gpu(dev):
    S.check_loss(dev)
