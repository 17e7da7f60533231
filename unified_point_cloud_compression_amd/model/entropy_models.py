"""Mean-scale hyperprior: counterpart of the reference's `model/entropy_models.py`.

Same module names (`h_a`, `h_s`, `scale_nn`, `rescale_nn`, `quant_nn`, `entropy_bottleneck`,
`gaussian_conditional`) and the same arithmetic as `MeanScaleHyperprior.compress/decompress/forward`
(`model/entropy_models.py:236-490`), on [N,C] rows:

  * every coordinate set is stored in canonical (b,x,y,z) order and the convolutions are deterministic
    (fixed summation order, no atomics), so the reference's `Sorted*` wrappers and `sort_tensor` /
    `sort_points` calls (`model/entropy_models.py:28-126,364-365,432-433`) are identities here;
  * LeakyReLU is fused into the producing convolution's epilogue;
  * scale bound, gain, table-index search, quantisation and likelihood are one kernel (`pcc_gauss_*`).

The entropy-coder boundary carries rANS byte strings (`strings = [[y_string], [z_string]]` as in the reference); the
coder is the per-channel GPU rANS of libpcc_hip by default (`entropy_coder="ans"`: single host stream in CompressAI's
byte layout; `"symbols"`: hand the int32 symbol tensors across, for kernel-level measurements).
"""
import os

import torch
import torch.nn as nn

from .. import MinkowskiEngine as ME
from .. import lib as L
from .. import sparse as S
from ..MinkowskiEngine.sparse_tensor import SparseTensor
from ..compressai.entropy_models import EntropyBottleneck, GaussianConditional
from ..compressai.models.base import CompressionModel
from ..compressai.ops import LowerBound


class SortedMinkowskiConvolution(ME.MinkowskiConvolution):
    """Kept for state-dict / API parity; canonical storage makes the sort a no-op."""


class SortedMinkowskiGenerativeConvolutionTranspose(ME.MinkowskiGenerativeConvolutionTranspose):
    pass


class SortedMinkowskiLeakyReLU(ME.MinkowskiLeakyReLU):
    pass


def _mlp(sizes, last=None):
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(nn.Linear(sizes[i], sizes[i + 1]))
        if i < len(sizes) - 2:
            layers.append(nn.ReLU())
    if last is not None:
        layers.append(last)
    return nn.Sequential(*layers)


class MeanScaleHyperprior(CompressionModel):
    def __init__(self, config):
        super().__init__()
        Cb, Ch = config["C_bottleneck"], config["C_hyper_bottleneck"]
        self.inverse_rescaling = config["inverse_rescaling"]
        self.quantization_mode = config["quantization_mode"]
        self.entropy_bottleneck_vbr = config["entropy_bottleneck_vbr"]
        self.adaptive_BN = config["adaptive_BN"] if "adaptive_BN" in config else True
        self.eps = 0.0001
        self.quantization_offset = config["quantization_offset"]
        self.entropy_coder = config.get("entropy_coder", "pcc_streams")
        if self.entropy_coder not in ("pcc_streams", "ans", "symbols"):
            raise L.PccError(f"unknown entropy_coder {self.entropy_coder!r}")
        coder = None if self.entropy_coder == "symbols" else self.entropy_coder
        if self.entropy_bottleneck_vbr:
            raise L.PccError("entropy_bottleneck_vbr: CompressAI's EntropyBottleneckVbr (reference `model/entropy_models.py:163-173`, "
                             "used by its training forward only, `:253,275-279`; false in every shipped configuration) is not built")
        self.gaussian_conditional = GaussianConditional(None, entropy_coder=coder)
        self.entropy_bottleneck = EntropyBottleneck(Ch, entropy_coder=coder)
        conv, gen = ME.MinkowskiConvolution, SortedMinkowskiGenerativeConvolutionTranspose
        self.h_a = nn.Sequential(
            conv(in_channels=Cb, out_channels=Ch, kernel_size=3, dimension=3),
            ME.MinkowskiLeakyReLU(inplace=False),
            conv(in_channels=Ch, out_channels=Ch, kernel_size=3, stride=2, dimension=3),
            ME.MinkowskiLeakyReLU(inplace=False),
            conv(in_channels=Ch, out_channels=Ch, kernel_size=3, stride=2, dimension=3),
        )
        self.h_s = nn.Sequential(
            gen(in_channels=Ch, out_channels=Ch, kernel_size=2, stride=2, bias=True, dimension=3),
            SortedMinkowskiLeakyReLU(inplace=False),
            gen(in_channels=Ch, out_channels=Cb * 3 // 2, kernel_size=2, stride=2, bias=True, dimension=3),
            SortedMinkowskiLeakyReLU(inplace=False),
            SortedMinkowskiConvolution(in_channels=Cb * 3 // 2, out_channels=Cb * 2, kernel_size=3, stride=1,
                                       dimension=3, bias=True),
        )
        self.scale_nn = _mlp([2, 8, Cb // 4, Cb], nn.Softplus())
        self.rescale_nn = _mlp([2, 8, Cb // 4, Cb], nn.Softplus())
        self.quant_nn = _mlp([2, 10, 10, 1])

    # ---- fused sub-networks (same arithmetic as the Sequentials above) -------------------------------
    @staticmethod
    def _conv_act(layer, x, act):
        cs = x._cset
        if isinstance(layer, ME.MinkowskiGenerativeConvolutionTranspose):
            out_set = cs.expand(layer.kernel_size, cs.ts // layer.stride)
            kmap = cs.csr_map(layer.kernel_size, cs.ts // layer.stride) or cs.kernel_map(
                out_set, layer.kernel_size, transposed=True, up_stride=layer.stride)
        else:
            out_set = cs if layer.stride == 1 else cs.stride(cs.ts * layer.stride)
            kmap = cs.kernel_map(out_set, layer.kernel_size)
        f = layer._apply_conv(x, out_set, kmap, act=act, slope=0.01)
        return SparseTensor._from_canonical(out_set, f)

    def hyper_analysis(self, y):
        x = self._conv_act(self.h_a[0], y, L.ACT_LEAKY)
        x = self._conv_act(self.h_a[2], x, L.ACT_LEAKY)
        return self._conv_act(self.h_a[4], x, L.ACT_NONE)

    def hyper_synthesis(self, z_hat):
        x = self._conv_act(self.h_s[0], z_hat, L.ACT_LEAKY)
        x = self._conv_act(self.h_s[2], x, L.ACT_LEAKY)
        return self._conv_act(self.h_s[4], x, L.ACT_NONE)

    def plan_synthesis(self, z_cset, y_cset):
        """Coordinate-only pre-pass of the hyper-synthesis (inference): its two generative expansions, their pair lists and the
        map of the last convolution onto y's rows depend on the COORDINATES of z and y alone, so the decoder queues them while
        the hyper-latent's rANS decode runs on the side stream instead of in front of the features that wait for them (~25
        small launches, no host read: a k2-s2 expansion makes exactly 8 children per row).  Everything is cached on the sets;
        `_gaussian_params` finds it.  (The encoder's opening is host-bound: queued there the same launches cost 0.15 ms.)"""
        if torch.is_grad_enabled() or not S.USE_GRID or z_cset.n == 0 or y_cset.n == 0:
            return
        cs = z_cset
        for layer in (self.h_s[0], self.h_s[2]):
            if not isinstance(layer, ME.MinkowskiGenerativeConvolutionTranspose) or cs.ts % layer.stride:
                return
            out_set = cs.expand(layer.kernel_size, cs.ts // layer.stride)
            if cs.csr_map(layer.kernel_size, cs.ts // layer.stride) is None:
                cs.kernel_map(out_set, layer.kernel_size, transposed=True, up_stride=layer.stride)
            cs = out_set
        last = self.h_s[4]
        if last.kernel_size == 3 and last.stride == 1 and cs.ts == y_cset.ts:
            cs.kernel_map(y_cset, 3)

    def gaussian_conditional_channels(self):
        return self.h_s[4].out_channels // 2

    def get_offsets(self, stddev, scale):
        """`quant_nn` on (scale, stddev) pairs per element (`model/entropy_models.py:218-233`)."""
        q = self.quant_nn
        if (torch.is_grad_enabled() and stddev.is_cuda and len(q) == 5 and q[0].in_features == 2 and q[0].out_features == 10
                and q[2].out_features == 10 and q[4].out_features == 1 and L.load().pcc_quant_mlp_params() == 151):
            from ..autograd import QuantMlpFn            # training: one kernel per direction instead of ~25 launches
            return QuantMlpFn.apply(scale.expand_as(stddev), stddev, q[0].weight, q[0].bias, q[2].weight, q[2].bias,
                                    q[4].weight, q[4].bias)
        return self.quant_nn(torch.stack([scale, stddev], dim=-1)).squeeze(-1)

    def _gains(self, q, y_cset, n_ch):
        """(scale rows [nb,C] | None, rescale rows [nb,C] | None) indexed by batch id
        (`model/entropy_models.py:386-393,451-465`)."""
        if not self.adaptive_BN:
            return None, None
        scale = self.scale_nn(q) + self.eps
        rescale = 1.0 / scale if self.inverse_rescaling else 1.0 / self.rescale_nn(q)
        return scale.to(torch.float32).contiguous(), rescale.to(torch.float32).contiguous()

    def _gaussian_params(self, z_hat, y_cset):
        """scales | means of y from the decoded hyper-latent (`h_s` + `features_at_coordinates`).  In the codec both sides
        must get the SAME BITS out of this (the scales pick the rANS table rows, `model/entropy_models.py:396-400,468-484`):
        its products always run in the six-term form -- one form, no range condition, hence no guard and no fallback that
        one side could take without the other (the scope is pinned: an enclosing fallback scope does not change it)."""
        if torch.is_grad_enabled():
            return self._gaussian_params_impl(z_hat, y_cset)
        with L.arith_scope(L.ARITH_BF6, pinned=True):
            return self._gaussian_params_impl(z_hat, y_cset)

    def _gaussian_params_impl(self, z_hat, y_cset):
        if not torch.is_grad_enabled() and S.USE_GRID:
            # the last h_s layer is only ever read at y's coordinates (`features_at_coordinates`, a6): evaluate the
            # 3x3x3 convolution for those rows alone (13 k of the 56 k octree children) instead of everywhere + gather
            x = self._conv_act(self.h_s[0], z_hat, L.ACT_LEAKY)
            x = self._conv_act(self.h_s[2], x, L.ACT_LEAKY)
            last, cs = self.h_s[4], x._cset
            if last.kernel_size == 3 and last.stride == 1 and cs.ts == y_cset.ts and y_cset.n > 0:
                kmap = cs.kernel_map(y_cset, 3)
                f = last._apply_conv(x, y_cset, kmap)
                centre = kmap.nbr[13 * y_cset.n:14 * y_cset.n]        # offset (0,0,0): is the row itself in the set?
                return torch.where((centre >= 0).unsqueeze(1), f, f.new_zeros(1))      # absent coordinate -> zeros
            g = self._conv_act(last, x, L.ACT_NONE)
            return S.lookup_gather(g._cset, g._canonical_features(), y_cset.keys, y_cset.n)
        g = self.hyper_synthesis(z_hat)
        gf = g._canonical_features()
        if torch.is_grad_enabled() and gf.requires_grad:       # training: differentiable row gather
            rows = torch.empty(max(y_cset.n, 1), dtype=torch.int32, device=gf.device)
            L.call("pcc_lookup_rows", L.ptr(g._cset.keys), g._cset.n, L.ptr(y_cset.keys), y_cset.n, L.ptr(rows), L.stream())
            rows = rows[:y_cset.n].long()
            return torch.where((rows >= 0).unsqueeze(1), gf[rows.clamp(min=0)], gf.new_zeros(1))
        return S.lookup_gather(g._cset, gf, y_cset.keys, y_cset.n)   # [Ny, 2C]

    # ---- reference API ---------------------------------------------------------------------------------
    def compress(self, y, q):
        """Returns (points, strings, shape) like the reference (`model/entropy_models.py:344-406`):
        points = [y.C, z.C], strings = [[y_string], [z_string]] (bytes), shape = [Nz].
        With entropy_coder="symbols" the strings are the int32 symbol tensors [y_symbols, z_symbols]."""
        z = self.hyper_analysis(y)
        z_sym, z_hat_f, _ = self.entropy_bottleneck.encode_rows(z._canonical_features(), want_likelihood=False)
        zj = self._start_z_streams(z_sym)
        z_hat = SparseTensor._from_canonical(z._cset, z_hat_f)
        params = self._gaussian_params(z_hat, y._cset)
        scale, _ = self._gains(q, y._cset, y.F.shape[1])
        y_sym, idx, _ = self.gaussian_conditional.encode_rows(y._canonical_features(), params, y._cset.keys, scale,
                                                              want_likelihood=False)
        if self.entropy_coder != "pcc_streams" and L.arith() == L.ARITH_H3:
            guard = L.h_guard(z_sym.device)               # (the stream coder reads the guard with its container header)
            if guard.item():
                guard.zero_()
                raise L.RangeGuardTripped()
        if self.entropy_coder == "symbols":
            return [y.C, z.C], [y_sym, z_sym], [z._cset.n]
        if self.entropy_coder == "pcc_streams":
            z_string, y_string = self._code_streams(z_sym, y_sym, idx, zj)
        else:
            z_string = self.entropy_bottleneck.compress_rows(z_sym)
            y_string = self.gaussian_conditional.compress_rows(y_sym, idx)
        return [y.C, z.C], [[y_string], [z_string]], [z._cset.n]

    def _start_z_streams(self, z_sym):
        """The hyper-latent's rANS encode (a serial recurrence per stream: 0.2 ms on a handful of CUs) started on the side stream
        as soon as its symbols exist, beside the hyper-synthesis and y's quantisation on the main stream.  Returns the job (with
        `.ready`, the event the main stream waits for before it touches the container), or None when nothing was started."""
        if self.entropy_coder != "pcc_streams" or not self.SIDE_STREAM_DECODE:
            return None
        zj = self.entropy_bottleneck.streams_job(z_sym)
        if zj.adaptive:                                   # its stream count needs a host read first: coded in `_code_streams`
            return zj
        dev = z_sym.device
        main, side = torch.cuda.current_stream(dev), L.side_stream(dev)
        side.wait_event(main.record_event())              # the symbols are on their way on the main stream
        with torch.cuda.stream(side):
            zj.launch_encode()
            zj.ready = side.record_event()
        zj.blob.record_stream(main)                       # allocated under the side stream, read on the main one
        z_sym.record_stream(side)
        return zj

    def _code_streams(self, z_sym, y_sym, idx, zj=None):
        """Both strings with two device->host reads: the hyper-latent's container comes back in one copy that also carries
        the payload estimate of y (which sizes y's stream count); y's container in the second."""
        started = zj is not None and getattr(zj, "ready", None) is not None
        if zj is None:
            zj = self.entropy_bottleneck.streams_job(z_sym)
        if started:
            torch.cuda.current_stream(z_sym.device).wait_event(zj.ready)
        yj = self.gaussian_conditional.streams_job(y_sym, idx)
        guard = L.h_guard(z_sym.device)                   # every matrix product of the encoder is queued by now: its range guard
        if zj.adaptive:                                   # large frames: the hyper-latent needs its own estimate first
            est = L.counter(2)
            zj.launch_estimate(est.data_ptr())
            if yj.adaptive:
                yj.launch_estimate(est.data_ptr() + 8)
            ez, ey = L.read(est)
            zj.launch_encode(ez)
            zj.attach(guard)
            yj.launch_encode(ey if yj.adaptive else None)
            z_string = zj.fetch()
        else:
            if not started:
                zj.launch_encode()
            zj.attach(guard)                              # ... travels in the header of the hyper-latent's container
            if yj.adaptive:
                yj.launch_estimate(zj.guest_ptr())
            else:
                yj.launch_encode()
            z_string = zj.fetch()
            if yj.adaptive:
                yj.launch_encode(zj.guest)
        if zj.guest2 and L.arith() == L.ARITH_H3:
            guard.zero_()
            raise L.RangeGuardTripped()
        return z_string, yj.fetch()

    def likelihoods(self, y, q):
        """Eval-mode likelihoods of y and z (what `forward` feeds the rate loss, `loss.py:63-81`)."""
        z = self.hyper_analysis(y)
        _, z_hat_f, z_lik = self.entropy_bottleneck.encode_rows(z._canonical_features())
        params = self._gaussian_params(SparseTensor._from_canonical(z._cset, z_hat_f), y._cset)
        scale, _ = self._gains(q, y._cset, y.F.shape[1])
        _, _, y_lik = self.gaussian_conditional.encode_rows(y._canonical_features(), params, y._cset.keys, scale)
        return y_lik, z_lik

    SIDE_STREAM_DECODE = os.environ.get("PCC_SIDE_DECODE", "1") != "0"

    def predecode(self, symbols, shape, device, check=None):
        """What depends on the strings alone, started before the decoder's coordinate work and beside it: the hyper-latent's
        rANS decode (a serial recurrence per stream on a handful of CUs: 0.4 ms during which the main stream builds y's and
        z's coordinate sets, the first level's maps and candidate set) and the upload of y's string, on the side stream.
        Returns a token for `decompress(pre=...)`, or None when the coder is not the GPU one."""
        if self.entropy_coder != "pcc_streams" or not self.SIDE_STREAM_DECODE or int(shape[0]) <= 0:
            return None
        (y_string,), (z_string,) = symbols
        eb = self.entropy_bottleneck
        eb._check_tables()
        status = L.counter(1, torch.int32)
        main = torch.cuda.current_stream(device)
        side = L.side_stream(device)
        side.wait_event(main.record_event())        # the counter block's zero fill (and whatever still uses a recycled
        #                                             allocation) is ordered on the main stream
        with torch.cuda.stream(side):
            z_sym = eb.decompress_rows(z_string, int(shape[0]), eb.channels, device=device, check=check, status=status)
            ev = side.record_event()
        z_sym.record_stream(main)                   # allocated under the side stream, consumed on the main one
        return z_sym, None, ev

    def predecode_upload(self, pre, symbols, device):
        """Second half of `predecode`, called once the decoder's coordinate work is queued: y's string goes to the device on
        the side stream (a pageable copy of ~1 MB keeps the host for ~0.1 ms -- time the main stream now spends on the
        coordinate sets while the side stream decodes the hyper-latent)."""
        if pre is None:
            return None
        z_sym, _, ev = pre
        main = torch.cuda.current_stream(device)
        side = L.side_stream(device, 1)             # (its own stream: behind the hyper-latent's decode the copy would wait for it)
        with torch.cuda.stream(side):
            y_up = self.gaussian_conditional.upload_string(symbols[0][0], device)
            ev2 = side.record_event()
        y_up.device_buf.record_stream(main)
        return z_sym, y_up, (ev, ev2)

    def decompress(self, points, symbols, shape, q, check=None, pre=None):
        """points = [y CoordSet, z CoordSet]; symbols = strings [[y_string], [z_string]] (or the symbol tensors with
        entropy_coder="symbols") (`model/entropy_models.py:409-490`).  Returns y_hat as a stride-8 SparseTensor."""
        assert isinstance(symbols, list) and len(symbols) == 2
        assert isinstance(points, list) and len(points) == 2
        y_cset, z_cset = points
        dev = y_cset.device
        y_ready = None
        c_y = self.gaussian_conditional_channels()
        if self.entropy_coder == "symbols":
            y_sym, z_sym = symbols
        else:
            (y_string,), (z_string,) = symbols
            if pre is not None:
                z_sym, y_up, evs = pre
                evs = evs if isinstance(evs, tuple) else (evs,)
                torch.cuda.current_stream(dev).wait_event(evs[0])                 # the hyper-latent's symbols
                if y_up is not None:
                    y_string, y_ready = y_up, evs[1]
            else:
                z_sym = self.entropy_bottleneck.decompress_rows(z_string, z_cset.n, self.entropy_bottleneck.channels,
                                                                device=dev, check=check)
            y_sym = None
        med = self.entropy_bottleneck.quantiles[:, 0, 1].detach().to(torch.float32)
        z_hat = SparseTensor._from_canonical(z_cset, z_sym.to(torch.float32) + med[None, :])
        params = self._gaussian_params(z_hat, y_cset)
        scale, rescale = self._gains(q, y_cset, c_y)
        if y_sym is None:
            idx = self.gaussian_conditional.index_rows(params, y_cset.keys, scale)
            if y_ready is not None:
                torch.cuda.current_stream(dev).wait_event(y_ready)               # y's string is on the device
            y_sym = self.gaussian_conditional.decompress_rows(y_string, y_cset.n, c_y, idx, check=check)
        if self.quantization_offset:
            c = y_sym.shape[1]
            scales_hat, means_hat = params[:, :c], params[:, c:]
            b = (y_cset.keys[:y_cset.n] >> 48)
            g = scale[b] if scale is not None else torch.ones_like(scales_hat)
            rg = rescale[b] if rescale is not None else torch.ones_like(scales_hat)
            qv = y_sym.to(torch.float32)
            q_abs, signs = qv.abs(), torch.sign(qv)
            stdev = self.gaussian_conditional.lower_bound_scale(scales_hat * g)
            off = -self.get_offsets(stdev, g)
            off[q_abs < 0.0001] = 0
            y_hat = signs * (q_abs + off) * rg + means_hat
        else:
            y_hat, _ = self.gaussian_conditional.decode_rows(y_sym, params, y_cset.keys, scale)
        return SparseTensor._from_canonical(y_cset, y_hat)

    noise_fn = None     # callable(tag, like) -> U(-.5,.5) noise; tests install a deterministic one

    def _noise(self, tag, like):
        if self.noise_fn is not None:
            return self.noise_fn(tag, like)
        return torch.empty_like(like).uniform_(-0.5, 0.5)

    def forward(self, y, q):
        """Training forward (`model/entropy_models.py:236-340`) on [N,C] rows: proxy quantisation (additive uniform
        noise / straight-through rounding), differentiable likelihoods, optional quantisation offsets.
        Returns (y_hat SparseTensor at stride 8, (y_likelihoods [Ny,C], z_likelihoods [Nz,Ch]))."""
        from ..compressai.ops.ops import quantize_ste
        z = self.hyper_analysis(y)
        zf, yf = z._canonical_features(), y._canonical_features()
        yb = y._cset.keys[:y._cset.n] >> 48
        scale, rescale = self._gains(q, y._cset, yf.shape[1])
        if scale is not None:
            # rows of the per-batch gains by batch index, as a one-hot product: the backward pass of `scale[yb]` is an
            # index_put with accumulation (a sort and ~10 launches, 0.29 ms per training step); this one is a [nb, n] x [n, C] GEMM
            hot = torch.nn.functional.one_hot(yb, scale.shape[0]).to(scale.dtype)
            rescale = (hot @ rescale.detach()) if self.inverse_rescaling else hot @ rescale
            scale = hot @ scale
        else:
            scale = rescale = torch.ones_like(yf)
        eb, gc = self.entropy_bottleneck, self.gaussian_conditional
        med = eb.quantiles[:, 0, 1].detach()
        nz = self._noise("z", zf)
        if self.quantization_mode == "uniform":
            z_hat_f = zf + nz
            z_lik = eb.likelihood_rows(z_hat_f)
        else:
            z_lik = eb.likelihood_rows(zf + nz)
            z_hat_f = quantize_ste(zf - med) + med
        params = self._gaussian_params(SparseTensor._from_canonical(z._cset, z_hat_f), y._cset)
        c = yf.shape[1]
        scales_hat, means_hat = params[:, :c], params[:, c:]
        ny = self._noise("y", yf)
        if self.quantization_offset:
            tmp = scale * (yf - means_hat)
            signs = torch.sign(tmp).detach()
            a = torch.abs(tmp)
            y_q_abs = a + ny if self.quantization_mode == "uniform" else quantize_ste(a)
            # the likelihood branch draws its own noise (`gaussian_conditional(...)` in training mode, reference
            # `model/entropy_models.py:312-316`), independent of `quantize_noise` at `:304`
            n_lik = self._noise("y_lik", yf) if self.quantization_mode == "uniform" else ny
            y_lik = gc.likelihood_rows(yf * scale + n_lik, scales_hat * scale, means_hat * scale)
            stdev = gc.lower_bound_scale(scales_hat * scale)
            off = -self.get_offsets(stdev, scale.detach())
            off = torch.where(y_q_abs < 0.0001, off.new_zeros(1), off)
            y_hat = signs * (y_q_abs + off) * rescale + means_hat
        else:
            y_t = yf * scale + ny
            y_lik = gc.likelihood_rows(y_t, scales_hat * scale, means_hat * scale)
            y_hat = y_t * rescale
        return SparseTensor._from_canonical(y._cset, y_hat), (y_lik, z_lik)
