"""refign_amd -- MI355X (gfx950) native implementation of Refign's align-and-refine hot path.

Host side mirrors the reference's operator/module interface for this path (brdav/refign):
  refign_amd.correlation   <-> models/correlation_ops  (pybind module `correlation` + spatial_correlation_sample)
  refign_amd.matching      <-> helpers/matching_utils.py (warp, mapping<->flow, confidence)
  refign_amd.modules       <-> models/modules.py (Local/GlobalFeatureCorrelationLayer, decoders, uncertainty)
  refign_amd.refine        <-> DomainAdaptationSegmentationModel.refine / .eta / .align tail

All compute goes through the C ABI of lib/librefign_hip.so (include/refign_hip.h).  There is NO CPU fallback:
calling an op without the HIP library or with CPU tensors raises.
"""
from ._lib import abi_version, library_path, load_library  # noqa: F401

__all__ = ["abi_version", "library_path", "load_library"]
