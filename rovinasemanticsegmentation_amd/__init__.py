"""rovinasemanticsegmentation_amd -- MI355X-native per-pixel RF + DenseCRF inference path.

Host-side mirror of the reference's Segmenter hot loop over librvseg.so (HIP, gfx950).
`import torch` before this package if the process also uses torch, so that both share one HIP
runtime.  There is no CPU fallback: without the built library the import of `_capi.lib()` fails,
without a GPU `Context()` raises.
"""
from . import _capi as capi  # noqa: F401
from .segmenter import Context, DenseCRF, FeatureExtractor, LocalMapStore, RandomForest, Segmenter  # noqa: F401
from . import synthetic  # noqa: F401

__all__ = ["capi", "Context", "DenseCRF", "FeatureExtractor", "LocalMapStore", "RandomForest", "Segmenter", "synthetic"]
