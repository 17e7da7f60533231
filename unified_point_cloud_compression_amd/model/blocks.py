"""`MinkowskiGDN` (reference `model/blocks.py:8-57`) as one fused HIP kernel.

norm = beta + |x| @ gamma^T ; y = x / norm (GDN) or x * norm (IGDN)  -- the GDN1 form, no square / sqrt
(`model/blocks.py:46`).  The reference transposes, calls conv1d, multiplies and re-hashes all coordinates into a
new SparseTensor; here the [C,C] product runs on the fp32 MFMA with the division fused into the epilogue and
the coordinate set is re-used.
"""
import torch

from .. import lib as L
from ..compressai.layers import GDN


class MinkowskiGDN(GDN):
    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1, kernel_size=1):
        super().__init__(in_channels, inverse, beta_min, gamma_init)
        self.kernel_size = kernel_size
        self.in_channels = int(in_channels)
        self._tag, self._packed, self._beta_eff = None, None, None

    def _pack(self):
        tag = (self.beta.data_ptr(), self.beta._version, self.gamma.data_ptr(), self.gamma._version)
        if tag != self._tag:
            c = self.in_channels
            dev = self.gamma.device
            n = L.load().pcc_gdn_packed_elems(c)
            if n <= 0:
                raise L.PccError(f"pcc_gdn: channel count {c} unsupported (needs 4, 8, 16 or a multiple of 32)")
            self._packed = torch.empty(n, dtype=torch.float32, device=dev)
            self._beta_eff = torch.empty(c, dtype=torch.float32, device=dev)
            L.call("pcc_gdn_pack", L.ptr(self.beta.detach().contiguous()), L.ptr(self.gamma.detach().contiguous()), c,
                   float(self.beta_min), L.ptr(self._packed), self._packed.numel(), L.ptr(self._beta_eff), L.stream())
            self._tag = tag
        return self._packed, self._beta_eff

    def forward_rows(self, feats):
        """[N,C] canonical-order rows -> GDN / IGDN rows."""
        if torch.is_grad_enabled() and (feats.requires_grad or self.gamma.requires_grad):
            from ..autograd import GdnFn            # training path (BASELINE config 4): fused forward, library-kernel backward
            return GdnFn.apply(feats, self.beta, self.gamma, self)
        feats = feats.contiguous()
        out = torch.empty_like(feats)
        packed, beta_eff = self._pack()
        L.call("pcc_gdn_fwd", L.ptr(feats), feats.shape[0], self.in_channels, L.ptr(packed), L.ptr(beta_eff),
               1 if self.inverse else 0, L.ptr(out), L.arith(), L.stream())
        return out

    def forward(self, x):
        return x._like(self.forward_rows(x._canonical_features()))
