"""`compressai.entropy_models` subset: `EntropyBottleneck`, `GaussianConditional`
(`model/entropy_models.py:8,161,175,272,282-285,312-319,371-372,396-400,438,468-484`).

Tensor layout at this surface is CompressAI's [B, C, N]; the HIP kernels work on row-major [N, C]
(the layout the sparse convolutions produce), so the [N, C] entry points `*_rows` are what the
build's own model uses and the [B, C, N] methods transpose around them.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from ... import lib as L
from ..ops import LowerBound

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(min=SCALES_MIN, max=SCALES_MAX, levels=SCALES_LEVELS):
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


def _rows(x):
    """[B,C,*] -> ([N,C] contiguous, restore fn)."""
    B, Cc = x.shape[0], x.shape[1]
    perm = x.reshape(B, Cc, -1).permute(0, 2, 1).contiguous()        # [B,N,C]
    n = perm.shape[1]

    def back(r):
        return r.reshape(B, n, Cc).permute(0, 2, 1).reshape(x.shape)
    return perm.reshape(B * n, Cc), back


class EntropyModel(nn.Module):
    def __init__(self, likelihood_bound=1e-9, entropy_coder=None, entropy_coder_precision=16):
        super().__init__()
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())

    @staticmethod
    def _no_coder():
        raise L.PccError("rANS byte coding is SURVEY 8f row 1 (next): this round hands integer symbols across the "
                         "entropy-coder boundary (`*_rows` APIs); compress()/decompress() to byte strings are not built")

    def compress(self, *a, **k):
        self._no_coder()

    def decompress(self, *a, **k):
        self._no_coder()


class EntropyBottleneck(EntropyModel):
    """Factorised prior.  Parameter names/shapes as CompressAI: `_matrix{i}`, `_bias{i}`,
    `_factor{i}`, `quantiles` [C,1,3] (`train.py:64-65` selects parameters ending in `.quantiles`)."""

    def __init__(self, channels, *args, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3), **kwargs):
        super().__init__(*args, **kwargs)
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        f = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / f[i + 1]))
            m = torch.Tensor(channels, f[i + 1], f[i])
            m.data.fill_(init)
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(m))
            b = torch.Tensor(channels, f[i + 1], 1)
            nn.init.uniform_(b, -0.5, 0.5)
            self.register_parameter(f"_bias{i:d}", nn.Parameter(b))
            if i < len(self.filters):
                fa = torch.Tensor(channels, f[i + 1], 1)
                nn.init.zeros_(fa)
                self.register_parameter(f"_factor{i:d}", nn.Parameter(fa))
        self.quantiles = nn.Parameter(torch.Tensor(channels, 1, 3))
        self.quantiles.data = torch.Tensor([-self.init_scale, 0, self.init_scale]).repeat(self.quantiles.size(0), 1, 1)
        self.register_buffer("target", torch.Tensor([np.log(2 / self.tail_mass - 1)]))
        self._packed_tag, self._packed = None, None

    def _get_medians(self):
        return self.quantiles[:, :, 1:2].detach()

    def _logits_cumulative(self, inputs, stop_gradient=False):
        logits = inputs
        for i in range(len(self.filters) + 1):
            m = getattr(self, f"_matrix{i:d}")
            b = getattr(self, f"_bias{i:d}")
            if stop_gradient:
                m, b = m.detach(), b.detach()
            logits = torch.matmul(torch.nn.functional.softplus(m), logits) + b
            if i < len(self.filters):
                fa = getattr(self, f"_factor{i:d}")
                if stop_gradient:
                    fa = fa.detach()
                logits = logits + torch.tanh(fa) * torch.tanh(logits)
        return logits

    def loss(self):
        logits = self._logits_cumulative(self.quantiles, stop_gradient=True)
        return torch.abs(logits - self.target).sum()

    def update(self, force=False):
        """Offsets / lengths of the per-channel tables; the quantised CDFs themselves belong to the rANS
        stage (8f row 1) and are built there."""
        if self._offset.numel() > 0 and not force:
            return False
        med = self.quantiles[:, 0, 1]
        minima = torch.ceil(med - self.quantiles[:, 0, 0]).int().clamp(min=0)
        maxima = torch.ceil(self.quantiles[:, 0, 2] - med).int().clamp(min=0)
        self._offset = -minima
        self._cdf_length = (maxima + minima + 1 + 2).int()
        return True

    def packed(self):
        """[C,58] = softplus(matrices) | biases | tanh(factors) for `pcc_eb_encode` (filters (3,3,3,3) only)."""
        if self.filters != (3, 3, 3, 3):
            raise L.PccError("pcc_eb_encode supports filters=(3,3,3,3) only")
        params = [getattr(self, f"_matrix{i}") for i in range(5)] + [getattr(self, f"_bias{i}") for i in range(5)] + \
                 [getattr(self, f"_factor{i}") for i in range(4)]
        tag = tuple((p.data_ptr(), p._version) for p in params)
        if tag != self._packed_tag:
            with torch.no_grad():
                parts = [torch.nn.functional.softplus(getattr(self, f"_matrix{i}")).reshape(self.channels, -1) for i in range(5)]
                parts += [getattr(self, f"_bias{i}").reshape(self.channels, -1) for i in range(5)]
                parts += [torch.tanh(getattr(self, f"_factor{i}")).reshape(self.channels, -1) for i in range(4)]
                self._packed = torch.cat(parts, dim=1).to(torch.float32).contiguous()
            assert self._packed.shape[1] == 58
            self._packed_tag = tag
        return self._packed

    def encode_rows(self, z, want_likelihood=True):
        """z [N,C] -> (symbols int32 [N,C], z_hat [N,C], likelihood [N,C] | None): eval-mode quantisation
        round(z - median) + median and its likelihood (`model/entropy_models.py:282-285,371-372`)."""
        z = z.contiguous()
        n, c = z.shape
        sym = torch.empty((n, c), dtype=torch.int32, device=z.device)
        zh = torch.empty_like(z)
        lik = torch.empty_like(z) if want_likelihood else None
        med = self.quantiles[:, 0, 1].detach().to(torch.float32).contiguous()
        L.call("pcc_eb_encode", L.ptr(z), n, c, L.ptr(self.packed()), L.ptr(med), L.ptr(sym), L.ptr(zh), L.ptr(lik),
               L.stream())
        return sym, zh, lik

    def forward(self, x, training=None):
        training = self.training if training is None else training
        if training:
            raise L.PccError("EntropyBottleneck noise-mode forward (training) is not built in this round")
        rows, back = _rows(x)
        _, zh, lik = self.encode_rows(rows)
        return back(zh), back(lik)


class GaussianConditional(EntropyModel):
    """Mean-scale Gaussian conditional."""

    def __init__(self, scale_table, *args, scale_bound=0.11, tail_mass=1e-9, **kwargs):
        super().__init__(*args, **kwargs)
        self.tail_mass = float(tail_mass)
        if scale_bound is None and scale_table:
            scale_bound = scale_table[0]
        self.lower_bound_scale = LowerBound(scale_bound)
        self.register_buffer("scale_table", torch.Tensor(tuple(float(s) for s in scale_table)) if scale_table else torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]) if scale_bound is not None else None)

    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        dev = self.scale_table.device
        self.scale_table = torch.as_tensor(scale_table, dtype=torch.float32).to(dev)
        self.update()
        return True

    def update(self):
        """pmf support of each scale (offset / length); quantised CDFs belong to the rANS stage (8f row 1)."""
        from scipy.stats import norm
        multiplier = -norm.ppf(self.tail_mass / 2)
        pmf_center = torch.ceil(self.scale_table.cpu() * multiplier).int()
        self._offset = (-pmf_center).to(self.scale_table.device)
        self._cdf_length = (2 * pmf_center + 1 + 2).to(self.scale_table.device)

    def _table(self, device):
        if self.scale_table.numel() == 0:
            raise L.PccError("GaussianConditional has no scale table: call model.update() first (`evaluate.py:89`)")
        return self.scale_table.to(device=device, dtype=torch.float32).contiguous()

    def encode_rows(self, y, params, keys=None, gain=None, want_likelihood=True, want_symbols=True):
        """Fused a7 kernel on [N,C] rows; params [N,2C] = (scales_hat | means_hat)."""
        y, params = y.contiguous(), params.contiguous()
        n, c = y.shape
        tab = self._table(y.device)
        sym = torch.empty((n, c), dtype=torch.int32, device=y.device) if want_symbols else None
        idx = torch.empty((n, c), dtype=torch.int32, device=y.device)
        lik = torch.empty_like(y) if want_likelihood else None
        g = gain.contiguous() if gain is not None else None
        L.call("pcc_gauss_encode", L.ptr(y), L.ptr(params), L.ptr(keys), L.ptr(g), n, c, L.ptr(tab), tab.numel(),
               L.ptr(sym), L.ptr(idx), L.ptr(lik), L.stream())
        return sym, idx, lik

    def decode_rows(self, sym, params, keys=None, gain=None):
        sym, params = sym.contiguous(), params.contiguous()
        n, c = sym.shape
        tab = self._table(sym.device)
        y_hat = torch.empty((n, c), dtype=torch.float32, device=sym.device)
        idx = torch.empty((n, c), dtype=torch.int32, device=sym.device)
        g = gain.contiguous() if gain is not None else None
        L.call("pcc_gauss_decode", L.ptr(sym), L.ptr(params), L.ptr(keys), L.ptr(g), n, c, L.ptr(tab), tab.numel(),
               L.ptr(y_hat), L.ptr(idx), L.stream())
        return y_hat, idx

    def build_indexes(self, scales):
        """[B,C,N] scales -> int32 indexes (`model/entropy_models.py:396,468`)."""
        rows, back = _rows(scales)
        params = torch.cat([rows, torch.zeros_like(rows)], dim=1)
        _, idx, _ = self.encode_rows(torch.zeros_like(rows), params, want_likelihood=False, want_symbols=False)
        return back(idx)

    def forward(self, inputs, scales, means=None, training=None):
        """Eval-mode forward: (round(x - mu) + mu, likelihood) (`model/entropy_models.py:312-316,329-333`)."""
        training = self.training if training is None else training
        if training:
            raise L.PccError("GaussianConditional noise-mode forward (training) is not built in this round")
        x, back = _rows(inputs)
        s, _ = _rows(scales)
        m = _rows(means)[0] if means is not None else torch.zeros_like(x)
        sym, _, lik = self.encode_rows(x, torch.cat([s, m], dim=1))
        return back(sym.to(torch.float32) + m), back(lik)
