"""refign_amd -- MI355X (gfx950) native implementation of Refign's align-and-refine hot path.

Host side mirrors the reference's operator/module interface for this path (brdav/refign):
  refign_amd.correlation   <-> models/correlation_ops  (pybind module `correlation` + spatial_correlation_sample)
  refign_amd.matching      <-> helpers/matching_utils.py (warp, mapping<->flow, confidence)
  refign_amd.modules       <-> models/modules.py (Local/GlobalFeatureCorrelationLayer, decoders, uncertainty)
  refign_amd.refine        <-> DomainAdaptationSegmentationModel.refine / .eta / .align tail

All compute of the hot path goes through the C ABI of lib/librefign_hip.so (include/refign_hip.h).  The kernel-backed operators
(correlation, warp, align tail, refine, L2 norm, uncertainty front end) have NO CPU form: they raise on CPU tensors or a missing
library.  The nn.Module mirrors of the segmentation networks (Linear, Conv2d, LayerNorm, attention ...) DO run on CPU tensors --
through ATen, for the host-side tests (configs, checkpoints, gloo data parallelism) -- and on a GPU hand any call outside their
kernels' domain to ATen too, recorded by mfma.note_library (bench.py prints `library_fallbacks`; the step goldens assert it empty).
"""
# GPU_MAX_HW_QUEUES (how many hardware queues ROCm multiplexes a process's streams onto; default 4) is left alone.
# Round 1 raised it to 8 for the eager step next to an RCCL communicator (328 -> 300 ms/step); with the student passes
# replayed from hipGraphs it is the other way round: next to an eagerly initialised communicator, 8 / 16 / 32 queues make
# the graph replays 1.5-1.7x slower (280-289 ms/step against 201.8 with 4; kernel durations unchanged, the three compute
# streams still run concurrently -- tools/micro/stream_probe.py), and without a communicator 4 / 8 / 16 are the same
# (199.7 / 200.9 / 200.2 ms).  profiles/r02_dist_1rank_queues.txt.

from ._lib import abi_version, library_path, load_library  # noqa: E402,F401

__all__ = ["abi_version", "library_path", "load_library"]
