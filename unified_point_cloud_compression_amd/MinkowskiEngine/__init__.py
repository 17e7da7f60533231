"""`MinkowskiEngine`-compatible operator surface backed by libpcc_hip (MI355X, gfx950).

Exposes exactly the symbols the reference touches, so that `model/transforms.py`
and `model/blocks.py` of ikt-luh/Unified-Point-Cloud-Compression load unchanged once
`unified_point_cloud_compression_amd.install_shims()` has registered this package as
`MinkowskiEngine`.  GPU only: there is no CPU fallback.
"""
from .sparse_tensor import SparseTensor  # noqa: F401
from .modules import (  # noqa: F401
    MinkowskiConvolution,
    MinkowskiGenerativeConvolutionTranspose,
    MinkowskiConvolutionTranspose,
    MinkowskiReLU,
    MinkowskiLeakyReLU,
    MinkowskiPruning,
    MinkowskiAvgPooling,
    MinkowskiChannelwiseConvolution,
)
from . import utils  # noqa: F401

__version__ = "0.5.4+pcc_hip"
