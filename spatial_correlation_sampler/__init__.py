"""Zero-patch plug-in point of the reference: `LocalFeatureCorrelationLayer.__init__` imports
`spatial_correlation_sample` from a package of THIS name before falling back to its own JIT-built op
(models/modules.py:252-262).  With this repo on PYTHONPATH the reference therefore runs on the MI355X kernel."""
from refign_amd.correlation import SpatialCorrelationSamplerFunction, spatial_correlation_sample  # noqa: F401
