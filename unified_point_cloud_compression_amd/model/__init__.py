"""Counterpart of the reference's `model/` package on the libpcc_hip operator surface."""
from .blocks import MinkowskiGDN  # noqa: F401
from .transforms import AnalysisTransform, SparseSynthesisTransform  # noqa: F401
from .entropy_models import MeanScaleHyperprior  # noqa: F401
from .model import UnifiedModel  # noqa: F401
