"""GPU: the evaluation forward -- DomainAdaptationSegmentationModel.forward in eval mode:
whole-image and sliding-window inference (batched and crop by crop), DAFormer and HRDA -- against goldens captured from
the reference (tests/golden/make_golden_eval.py; segmentation_model.py:304-382), plus validation_step -> IoU."""
import numpy as np
import pytest
import torch
from conftest import golden
from fill import hashed_uniform
from test_step_gpu import build

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@torch.no_grad()
@pytest.mark.parametrize("use_hrda,name", [(False, "eval_daformer_96x160"), (True, "eval_hrda_96x160")])
def test_eval_forward_matches_reference(dev, use_hrda, name):
    z = golden(name)
    H, W = [int(v) for v in z["size"]]
    x = torch.from_numpy((hashed_uniform((2, 3, H, W), "g15/img") * 4 - 2).astype(np.float32)).to(dev)
    model = build(use_hrda, dev).eval()
    outs = {}
    model.use_slide_inference = False
    outs["whole"] = model(x)
    model.use_slide_inference = True
    model.inference_crop_size, model.inference_stride = [64, 64], [40, 48]
    model.inference_batched_slide = True
    outs["slide_batched"] = model(x, out_size=(120, 200))
    model.inference_batched_slide = False
    serial = model(x, out_size=(120, 200))
    assert float((serial - outs["slide_batched"]).abs().max()) < 1e-3
    for k, v in outs.items():
        v = v.float().cpu().numpy()
        scale = np.abs(z[k + "_sample"]).max()
        assert np.abs(v[:, :, ::3, ::3] - z[k + "_sample"]).max() <= 2e-3 * scale, k
        assert abs(np.abs(v.astype(np.float64)).sum() - float(z[k + "_abs_checksum"])) <= 1e-3 * float(z[k + "_abs_checksum"])
        decided = z[k + "_margin"].astype(np.float32) > 20 * 2e-3 * scale
        assert decided.mean() > 0.5
        assert np.array_equal(v.argmax(1)[decided], z[k + "_argmax"][decided]), k


@torch.no_grad()
def test_validation_step_accumulates_iou(dev):
    from refign_amd.metrics import IoU, MyMetricCollection
    model = build(False, dev).eval()
    model.valid_metrics = MyMetricCollection({"val_ACDC_IoU": IoU(num_classes=19, ignore_index=255).to(dev),
                                              "val_DarkZurich_IoU": IoU(num_classes=19, ignore_index=255).to(dev)})
    x = torch.from_numpy((hashed_uniform((2, 3, 96, 160), "g15/img") * 4 - 2).astype(np.float32)).to(dev)
    y = torch.randint(0, 19, (2, 120, 200), device=dev)
    y[:, :10] = 255
    y_hat = model.validation_step({"image": x, "semantic": y}, 0, 0, src_name="DarkZurich")
    assert tuple(y_hat.shape) == (2, 19, 120, 200)
    pred = y_hat.argmax(1)
    keep = y != 255
    inter = torch.stack([((pred == c) & (y == c) & keep).sum() for c in range(19)]).float()
    union = torch.stack([(((pred == c) | (y == c)) & keep).sum() for c in range(19)]).float()
    want = torch.where(union > 0, inter / union.clamp(min=1), torch.zeros_like(union)).mean()
    out = model.validation_epoch_end()
    assert abs(float(out["val_DarkZurich_IoU"]) - float(want)) < 1e-6
    assert int(model.valid_metrics["val_ACDC_IoU"].confmat.sum()) == 0          # other dataset's metric untouched
    assert int(model.valid_metrics["val_DarkZurich_IoU"].confmat.sum()) == 0    # reset at epoch end


@torch.no_grad()
def test_predict_step_writes_label_and_colour_pngs(dev, tmp_path):
    from PIL import Image
    model = build(False, dev).eval()
    x = torch.from_numpy((hashed_uniform((2, 3, 96, 160), "g15/img") * 4 - 2).astype(np.float32)).to(dev)
    preds = model.predict_step({"image": x, "filename": ["a.png", "b.png"]}, save_dir=str(tmp_path), orig_size=(120, 200))
    assert preds.shape == (2, 120, 200) and preds.dtype == np.uint8
    want = model(x, out_size=(120, 200)).argmax(1).cpu().numpy()
    for i, name in enumerate(("a.png", "b.png")):
        ids = np.array(Image.open(tmp_path / "preds" / name))
        assert np.array_equal(ids, want[i])
        col = Image.open(tmp_path / "color_preds" / name)
        assert col.mode == "P" and np.array_equal(np.array(col), want[i])
        rgb = np.array(col.convert("RGB"))
        if (want[i] == 0).any():
            assert tuple(rgb[want[i] == 0][0]) == (128, 64, 128)       # road
