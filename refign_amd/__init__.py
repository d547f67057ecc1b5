"""refign_amd -- MI355X (gfx950) native implementation of Refign's align-and-refine hot path.

Host side mirrors the reference's operator/module interface for this path (brdav/refign):
  refign_amd.correlation   <-> models/correlation_ops  (pybind module `correlation` + spatial_correlation_sample)
  refign_amd.matching      <-> helpers/matching_utils.py (warp, mapping<->flow, confidence)
  refign_amd.modules       <-> models/modules.py (Local/GlobalFeatureCorrelationLayer, decoders, uncertainty)
  refign_amd.refine        <-> DomainAdaptationSegmentationModel.refine / .eta / .align tail

All compute goes through the C ABI of lib/librefign_hip.so (include/refign_hip.h).  There is NO CPU fallback:
calling an op without the HIP library or with CPU tensors raises.
"""
import os as _os

# The step runs the gradient-free teacher branch on a side stream next to the student forward/backward
# (refign_amd/uda.py).  ROCm multiplexes a process's streams onto 4 hardware queues by default and streams that share a
# queue do not overlap; an RCCL communicator takes several.  8 queues keeps the two compute streams apart (measured on
# MI355X with a 1-rank process group: 328 -> 300 ms/step).  Read by the HIP runtime when it initialises, i.e. at the
# first device call -- importing this package before that is enough; an explicit setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._lib import abi_version, library_path, load_library  # noqa: E402,F401

__all__ = ["abi_version", "library_path", "load_library"]
