"""`compressai.ops.ops.quantize_ste` (`model/entropy_models.py:285,308`)."""
import torch


def quantize_ste(x):
    """round(x) with a straight-through gradient."""
    return x + (torch.round(x) - x).detach()
