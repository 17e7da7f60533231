"""`ME.Minkowski*` module counterparts.

Constructor signatures, parameter names and shapes follow MinkowskiEngine 0.5.x so the reference's
`state_dict`s load: `kernel` is [K, Cin, Cout] ([Cin, Cout] when K == 1), `bias` is [1, Cout]
(SURVEY A.4).  Forward passes call libpcc_hip; under autograd they go through `autograd.SparseConvFn`.
"""
import math

import torch
import torch.nn as nn

from .. import lib as L
from .. import sparse as S
from .sparse_tensor import SparseTensor


def _iso(v, what):
    if isinstance(v, (list, tuple)):
        if len(set(int(x) for x in v)) != 1:
            raise L.PccError(f"anisotropic {what} {v} is not supported")
        return int(v[0])
    return int(v)


def _no_grad_guard(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        # parameters require grad by default; only an actual autograd graph is a problem
        pass


class _ConvBase(nn.Module):
    """Shared parameter handling of the dense-kernel sparse convolutions."""

    TRANSPOSED = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=None):
        super().__init__()
        if dimension != 3:
            raise L.PccError("only dimension=3 is supported")
        if kernel_generator is not None:
            raise L.PccError("custom kernel generators are not supported")
        if _iso(dilation, "dilation") != 1:
            raise L.PccError("dilation != 1 is not supported")
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.kernel_size = _iso(kernel_size, "kernel_size")
        self.stride = _iso(stride, "stride")
        self.dimension = 3
        K = self.kernel_size ** 3
        self.kernel_volume = K
        shape = (K, self.in_channels, self.out_channels) if K > 1 else (self.in_channels, self.out_channels)
        self.kernel = nn.Parameter(torch.empty(shape, dtype=torch.float32))
        self.bias = nn.Parameter(torch.empty(1, self.out_channels, dtype=torch.float32)) if bias else None
        self._packed = S.PackedConv(transposed=self.TRANSPOSED)
        self.reset_parameters()

    def reset_parameters(self):
        # MinkowskiEngine: uniform(-a, a), a = 1/sqrt(fan * K); fan = Cout for transposed convs
        fan = self.out_channels if self.TRANSPOSED else self.in_channels
        a = 1.0 / math.sqrt(fan * self.kernel_volume)
        with torch.no_grad():
            self.kernel.uniform_(-a, a)
            if self.bias is not None:
                self.bias.uniform_(-a, a)

    def _apply_conv(self, x, out_set, kmap, act=L.ACT_NONE, slope=0.01):
        feats = x._canonical_features()
        if torch.is_grad_enabled() and (feats.requires_grad or self.kernel.requires_grad):
            from ..autograd import SparseConvFn            # training path (BASELINE config 4)
            perm = S.weight_offset_perm(self.kernel_volume, self.kernel.device)
            kernel = self.kernel if perm is None else self.kernel[perm]
            return SparseConvFn.apply(feats, kernel, self.bias, self, x._cset, out_set, kmap, act, slope)
        packed = self._packed.get(self.kernel, state_dict_order=True)
        if isinstance(kmap, tuple):          # CSR pair lists from the fused coordinate expansion
            return S.convt_forward_csr(x._canonical_features(), packed, self.bias, self.kernel_volume,
                                       self.in_channels, self.out_channels, kmap, out_set.n, act, slope)
        fwd = S.convt_forward if self.TRANSPOSED else S.conv_forward
        return fwd(x._canonical_features(), packed, self.bias, self.kernel_volume, self.in_channels,
                   self.out_channels, kmap, out_set.n, act, slope)

    def extra_repr(self):
        return (f"in={self.in_channels}, out={self.out_channels}, kernel_size={self.kernel_size}, "
                f"stride={self.stride}, bias={self.bias is not None}")


class MinkowskiConvolution(_ConvBase):
    """`ME.MinkowskiConvolution` (17 reference sites: `model/transforms.py:33-43,127,142-166`,
    `model/entropy_models.py:178-182,190`)."""

    def forward(self, input, coordinates=None):
        if coordinates is not None:
            raise L.PccError("explicit output coordinates are not supported")
        cs = input._cset
        if self.stride == 1:
            out_set = cs
        else:
            out_set = cs.stride(cs.ts * self.stride)
        kmap = None if self.kernel_volume == 1 and self.stride == 1 else cs.kernel_map(out_set, self.kernel_size)
        out = self._apply_conv(input, out_set, kmap)
        if self.stride == 1:
            return input._like(out)
        return SparseTensor._from_canonical(out_set, out)


class MinkowskiGenerativeConvolutionTranspose(_ConvBase):
    """`ME.MinkowskiGenerativeConvolutionTranspose` (`model/transforms.py:129,133,137`,
    `model/entropy_models.py:186,188`): output support = union of all kernel offsets around every input."""

    TRANSPOSED = True

    def forward(self, input, coordinates=None):
        if coordinates is not None:
            raise L.PccError("explicit output coordinates are not supported")
        cs = input._cset
        if cs.ts % self.stride != 0:
            raise L.PccError(f"tensor_stride {cs.ts} not divisible by up-sampling stride {self.stride}")
        ts_out = cs.ts // self.stride
        out_set = cs.expand(self.kernel_size, ts_out)
        kmap = cs.csr_map(self.kernel_size, ts_out) or cs.kernel_map(out_set, self.kernel_size, transposed=True,
                                                                    up_stride=self.stride)
        out = self._apply_conv(input, out_set, kmap)
        return SparseTensor._from_canonical(out_set, out)


MinkowskiConvolutionTranspose = MinkowskiGenerativeConvolutionTranspose


class _Elementwise(nn.Module):
    def _run(self, input, fn):
        return input._like(fn(input._canonical_features()))


class MinkowskiReLU(_Elementwise):
    """`ME.MinkowskiReLU` (`model/transforms.py:148,153,158`)."""

    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, input):
        return self._run(input, torch.relu)


class MinkowskiLeakyReLU(_Elementwise):
    """`ME.MinkowskiLeakyReLU` (`model/entropy_models.py:95,179,181,187,189`)."""

    def __init__(self, negative_slope=0.01, inplace=False):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return self._run(input, lambda f: torch.nn.functional.leaky_relu(f, self.negative_slope))


class MinkowskiPruning(nn.Module):
    """`ME.MinkowskiPruning` (`model/transforms.py:163,280`): keep rows where mask is True, in the input's relative
    order; an all-False mask gives an empty tensor."""

    def forward(self, input, mask):
        if mask.dtype != torch.bool or mask.shape[0] != len(input):
            raise L.PccError("pruning mask must be a bool tensor with one entry per row")
        mask = mask.to(input.device)
        cs = input._cset
        if input._perm is None:
            keys, feats, k = S.prune(cs.keys, cs.n, input._canonical_features(), mask)
            return SparseTensor._from_canonical(S.CoordSet(keys, k, cs.ts, cs.bounds), feats)
        # user-ordered rows: the kept set in canonical order for the kernels, .C / .F in the caller's row order
        keys, _, k = S.prune(cs.keys, cs.n, None, mask[input._perm])
        out_set = S.CoordSet(keys, k, cs.ts, cs.bounds)
        rank = torch.empty(cs.n, dtype=torch.int64, device=input.device)      # user row -> canonical position
        rank[input._perm] = torch.arange(cs.n, device=input.device)
        kept_user = torch.nonzero(mask)[:, 0]
        order = torch.argsort(rank[kept_user])                                 # canonical position among kept -> kept user row
        t = SparseTensor.__new__(SparseTensor)
        C = input.C[kept_user].contiguous()
        C._pcc_cset, C._pcc_perm, C._pcc_version = out_set, order, C._version
        t._cset, t._perm, t._C, t._F, t._Fc = out_set, order, C, input.F[kept_user], None
        return t


class MinkowskiAvgPooling(nn.Module):
    """Constructed by `loss.py:124-125` but never called; forward is not part of the hot path."""

    def __init__(self, kernel_size, stride=1, dilation=1, kernel_generator=None, dimension=None):
        super().__init__()
        self.kernel_size, self.stride, self.dimension = kernel_size, stride, dimension

    def forward(self, input):
        raise L.PccError("MinkowskiAvgPooling.forward is outside the hot path (never called by the reference)")


class MinkowskiChannelwiseConvolution(nn.Module):
    """Constructed by the Shepard's-loss ablation (`loss.py:185`); out of scope."""

    def __init__(self, in_channels, kernel_size=-1, stride=1, dilation=1, bias=False, kernel_generator=None,
                 dimension=None):
        super().__init__()
        K = _iso(kernel_size, "kernel_size") ** 3
        self.kernel = nn.Parameter(torch.zeros(K, int(in_channels)))

    def forward(self, input):
        raise L.PccError("MinkowskiChannelwiseConvolution.forward is outside the hot path")
