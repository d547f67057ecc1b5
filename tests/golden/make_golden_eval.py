"""tests/golden/make_golden_eval.py -- G15: the evaluation forward of the reference:
`DomainAdaptationSegmentationModel.forward` in eval mode -- sliding-window inference (batched and one crop at a time) and
whole-image inference, DAFormer and HRDA -- on a small synthetic batch with closed-form weights.
    python tests/golden/make_golden_eval.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_import as R  # noqa: E402

R.setup()
from fill import hashed_uniform  # noqa: E402
from make_golden_step import run_reference, save  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    b, H, W = 2, 96, 160
    x = (hashed_uniform((b, 3, H, W), "g15/img") * 4 - 2).astype(np.float32)
    xt = torch.from_numpy(x)
    for use_hrda in (False, True):
        model = run_reference(use_hrda)
        model.eval()
        arrays = {"size": np.array([H, W])}
        model.use_slide_inference = False
        arrays["whole"] = model.forward(xt).numpy()
        model.use_slide_inference = True
        model.inference_crop_size, model.inference_stride = [64, 64], [40, 48]
        for batched in (True, False):
            model.inference_batched_slide = batched
            arrays["slide_batched" if batched else "slide_serial"] = model.forward(xt, out_size=(120, 200)).numpy()
        d = float(np.abs(arrays["slide_batched"] - arrays["slide_serial"]).max())
        assert d < 1e-4, d                                # same crops, batched or one by one (rounding only)
        small = {"size": arrays["size"], "serial_vs_batched_max_diff": np.float64(d)}
        for k in ("whole", "slide_batched"):            # fixtures stay small: every third pixel + checksums + argmax
            v = arrays[k]
            small[k + "_sample"] = v[:, :, ::3, ::3]
            small[k + "_abs_checksum"] = np.float64(np.abs(v.astype(np.float64)).sum())
            small[k + "_argmax"] = v.argmax(1).astype(np.uint8)
            srt = np.sort(v, axis=1)
            small[k + "_margin"] = (srt[:, -1] - srt[:, -2]).astype(np.float16)
        save("eval_hrda_96x160" if use_hrda else "eval_daformer_96x160", **small)
        print(use_hrda, {k: float(np.abs(v).mean()) for k, v in arrays.items() if k != "size"})


if __name__ == "__main__":
    main()
