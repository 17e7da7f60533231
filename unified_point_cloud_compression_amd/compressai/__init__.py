"""`compressai`-compatible surface (CompressAI 1.2.4 subset the reference imports).

Likelihood / quantisation arithmetic runs in libpcc_hip (`pcc_gauss_*`, `pcc_eb_encode`); parameter
containers, names and shapes follow CompressAI so reference `state_dict`s load.
"""
from . import layers, ops, models, entropy_models  # noqa: F401

__version__ = "1.2.4+pcc_hip"
