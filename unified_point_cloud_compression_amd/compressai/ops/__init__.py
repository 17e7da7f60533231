"""`compressai.ops`: `LowerBound`, `ops.quantize_ste` (`model/entropy_models.py:6`), `NonNegativeParametrizer`."""
import torch
import torch.nn as nn

from . import ops  # noqa: F401  (module `compressai.ops.ops` holding quantize_ste)
from .ops import quantize_ste  # noqa: F401


class _LowerBoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        pass_through = (x >= bound) | (g < 0)
        return pass_through.type(g.dtype) * g, None


class LowerBound(nn.Module):
    """max(x, bound) with the CompressAI gradient rule."""

    def __init__(self, bound):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return _LowerBoundFn.apply(x, self.bound)


class NonNegativeParametrizer(nn.Module):
    """reparam(x) = max(x, sqrt(minimum + 2^-36))^2 - 2^-36."""

    def __init__(self, minimum=0, reparam_offset=2 ** -18):
        super().__init__()
        self.minimum = float(minimum)
        self.reparam_offset = float(reparam_offset)
        pedestal = self.reparam_offset ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        bound = (self.minimum + self.reparam_offset ** 2) ** 0.5
        self.lower_bound = LowerBound(bound)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        out = self.lower_bound(x)
        return out ** 2 - self.pedestal
