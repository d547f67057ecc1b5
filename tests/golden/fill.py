"""tests/golden/fill.py -- deterministic, platform-exact test data (integer hashing only, no libm):

  hashed_uniform(shape, key)  -> float32 array in [0,1), a pure function of (key, flat index)
  closed_form_fill(module)    -> fills every parameter / buffer of a torch module as a pure function of its
                                 state_dict key, with magnitudes that keep activations O(1) through deep stacks.

Used by make_golden.py (with the reference's modules) and by the tests (with refign_amd's modules): because both sides
use identical state_dict keys, identical weights result, and fixtures only need to hold inputs and outputs.
"""
import zlib

import numpy as np


def _mix(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(33)
    x = (x * np.uint64(0xff51afd7ed558ccd)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(33)
    x = (x * np.uint64(0xc4ceb9fe1a85ec53)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(33)
    return x


def hashed_uniform(shape, key):
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(key.encode()) + 0x9E3779B9)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + seed * np.uint64(0xD1B54A32D192ED03)
        bits = _mix(idx) >> np.uint64(40)               # 24 random bits
    return (bits.astype(np.float32) / np.float32(1 << 24)).reshape(shape)


def closed_form_fill(module, prefix=""):
    """In-place fill of module.state_dict() tensors; returns the module."""
    import torch
    sd = module.state_dict()
    with torch.no_grad():
        for name, tsr in sd.items():
            if not tsr.is_floating_point():
                continue                                  # num_batches_tracked
            u = hashed_uniform(tuple(tsr.shape), prefix + name)
            c = u * 2.0 - 1.0                             # centred, var 1/3
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "running_var":
                v = 0.5 + u                               # [0.5, 1.5)
            elif leaf == "running_mean":
                v = 0.2 * c
            elif leaf == "bias":
                v = 0.1 * c
            elif tsr.dim() <= 1:                          # BN / LN gamma, layer-scale
                v = 1.0 + 0.2 * c
            else:                                         # conv / linear weight: He-like, var = 2 / fan_in
                fan_in = tsr[0].numel()
                v = c * np.sqrt(6.0 / fan_in)
            # regression heads of the matcher get small weights so that synthetic flows stay a few pixels (in-image)
            if any(h in name for h in ("predict_mapping.", "dc_convs.6.", "predict_uncertainty_final.")):
                v = 0.05 * v
            tsr.copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(tsr.dtype))
    return module
