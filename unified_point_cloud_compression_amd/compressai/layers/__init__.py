"""`compressai.layers.GDN` parameter container (`model/blocks.py:5,8,22,40-41`)."""
import torch
import torch.nn as nn

from ..ops import NonNegativeParametrizer


class GDN(nn.Module):
    """Holds `beta` [C], `gamma` [C,C] and their reparametrisers exactly as CompressAI does; the dense
    image forward is not needed by the reference (it subclasses and overrides `forward`)."""

    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        beta_min, gamma_init = float(beta_min), float(gamma_init)
        self.inverse = bool(inverse)
        self.beta_min = beta_min
        self.beta_reparam = NonNegativeParametrizer(minimum=beta_min)
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(gamma_init * torch.eye(in_channels)))

    def forward(self, x):
        _, C, _, _ = x.size()
        beta = self.beta_reparam(self.beta)
        gamma = self.gamma_reparam(self.gamma).reshape(C, C, 1, 1)
        norm = torch.nn.functional.conv2d(x ** 2, gamma, beta)
        norm = torch.sqrt(norm) if self.inverse else torch.rsqrt(norm)
        return x * norm
