"""`compressai.entropy_models` subset: `EntropyBottleneck`, `GaussianConditional`
(`model/entropy_models.py:8,161,175,272,282-285,312-319,371-372,396-400,438,468-484`).

Attribution: CompressAI (InterDigital, BSD 3-Clause Clear) is not part of /root/reference; parameter names, shapes,
initial values and the table construction (`update`: minima / maxima / pmf_start / pmf_length / samples) restate its
published `compressai/entropy_models/entropy_models.py` 1.2.4 because they ARE the state_dict schema and the bitstream
tables a drop-in must reproduce.  The arithmetic on the hot path (likelihood, index, quantise, rANS) is libpcc_hip.

Tensor layout at this surface is CompressAI's [B, C, N]; the HIP kernels work on row-major [N, C]
(the layout the sparse convolutions produce), so the [N, C] entry points `*_rows` are what the
build's own model uses and the [B, C, N] methods transpose around them.
"""
import ctypes as C
import math

import warnings

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


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def pmf_to_quantized_cdf(pmf, precision=16):
    """CompressAI `pmf_to_quantized_cdf` (C++ op) -> libpcc_hip host function."""
    pmf = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    out = np.zeros(len(pmf) + 1, dtype=np.int32)
    L.check(L.load().pcc_pmf_to_quantized_cdf(_np_ptr(pmf), len(pmf), precision, _np_ptr(out)), "pcc_pmf_to_quantized_cdf")
    return out


class EntropyModel(nn.Module):
    """Table container + entropy coding entry points.

    `entropy_coder`:
      "pcc_streams" (default)  GPU rANS, one stream per channel behind a length table (`pcc_rans_*_streams`)
      "ans"                    single stream on the host in CompressAI's `BufferedRansEncoder` byte layout
    Both use the same tables and the same rans64 arithmetic; only the container differs.
    """

    def __init__(self, likelihood_bound=1e-9, entropy_coder=None, entropy_coder_precision=16):
        super().__init__()
        self.entropy_coder = entropy_coder or "pcc_streams"
        if self.entropy_coder not in ("pcc_streams", "ans"):
            raise L.PccError(f"unknown entropy coder {self.entropy_coder!r}")
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        self._likelihood_bound_value = float(likelihood_bound)
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        self._tables_gen = 0          # bumped whenever the tables change: keys the packed encoder / decoder tables

    _RESIZABLE = ("_offset", "_quantized_cdf", "_cdf_length", "scale_table")

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        """A checkpoint written after `model.update()` (`train.py:169-174,322`) holds populated tables while a fresh
        model's are empty: resize the table buffers to the checkpoint's shapes before the copy, as CompressAI's
        `update_registered_buffers` does for `load_state_dict` (`evaluate.py:86`)."""
        for name in self._RESIZABLE:
            buf = self._buffers.get(name)
            src = state_dict.get(prefix + name)
            if buf is not None and src is not None and tuple(buf.shape) != tuple(src.shape):
                self._buffers[name] = torch.empty(tuple(src.shape), dtype=buf.dtype, device=buf.device)
        self._tables_gen += 1
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    # ---- tables -----------------------------------------------------------------------------------------
    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        """CompressAI `EntropyModel._pmf_to_cdf`: per row quantise [pmf[:length], tail] to 16-bit frequencies."""
        pmf, tail = pmf.detach().cpu().float().numpy(), tail_mass.detach().cpu().float().numpy().reshape(len(pmf), -1)
        length = pmf_length.detach().cpu().numpy()
        cdf = np.zeros((len(length), int(max_length) + 2), dtype=np.int32)
        for i in range(len(length)):
            c = pmf_to_quantized_cdf(np.concatenate([pmf[i, :length[i]], tail[i, :1]]), self.entropy_coder_precision)
            cdf[i, :len(c)] = c
        return torch.from_numpy(cdf)

    def _check_tables(self):
        if self._quantized_cdf.numel() == 0 or self._cdf_length.numel() == 0 or self._offset.numel() == 0:
            raise L.PccError("entropy tables are empty: call model.update() first (`evaluate.py:89`)")

    STREAM_SYMBOLS = 1536    # channel grouping: merge channels until a stream holds this many symbols
    SEGMENT_SYMBOLS = 4096   # row segmentation: cut streams longer than twice this (12 bytes of framing per stream)
    MAX_STREAMS = 4096

    def n_streams(self, n, c):
        """(channel groups, row segments) of the GPU streams for an [n, c] symbol matrix.  Small inputs: a power-of-two
        merge of channels so that every stream keeps >= STREAM_SYMBOLS symbols (rate).  Large inputs: one group per
        channel, rows cut into segments of >= SEGMENT_SYMBOLS symbols (speed: one GPU lane per stream).  Encoder and
        decoder both derive it from (n, c), so it is not transmitted."""
        ng = c
        while ng % 2 == 0 and n * (c // ng) < self.STREAM_SYMBOLS:
            g = c // (ng // 2)
            if g & (g - 1):            # group size must stay a power of two
                break
            ng //= 2
        per = n * (c // ng)
        segs = max(1, min(per // max(1, self.SEGMENT_SYMBOLS), max(1, self.MAX_STREAMS // ng)))
        return ng, segs

    # ---- coding of [N, C] int32 symbol rows (the layout the kernels produce) -------------------------------
    def compress_rows(self, sym, idx=None):
        """sym [N,C] int32 (device) with table rows idx [N,C] (None: row = channel) -> bytes."""
        self._check_tables()
        sym = sym.contiguous()
        n, c = sym.shape
        dev = sym.device
        cdf, sizes, offs = (t.to(dev).contiguous() for t in (self._quantized_cdf, self._cdf_length, self._offset))
        if self.entropy_coder == "ans":
            s = sym.t().contiguous().cpu().numpy().reshape(-1)                      # channel-major, as [1,C,N].reshape(-1)
            i = (idx.t().contiguous().cpu().numpy().reshape(-1) if idx is not None
                 else np.repeat(np.arange(c, dtype=np.int32), n))
            return self._host_encode(s, np.ascontiguousarray(i, np.int32))
        lib = L.load()
        ng, segs = self.n_streams(n, c)
        if self.FRAMING_TARGET > 0 and n * c >= self.ADAPTIVE_MIN_SYMBOLS:
            est = L.counter()
            L.call("pcc_rans_estimate_bits", L.ptr(sym), L.ptr(idx.contiguous()) if idx is not None else None, n, c, L.ptr(cdf),
                   cdf.shape[1], L.ptr(sizes), L.ptr(offs), L.cptr(est), L.stream())
            segs = self._adaptive_segments(n, c, ng, segs, int(L.read(est)[0]) / 2048.0)
        ns = ng * segs
        per = lib.pcc_rans_stream_symbols(n, c, ng, segs)
        cap = lib.pcc_rans_container_max_bytes(per, ns)
        out = torch.empty(cap, dtype=torch.uint8, device=dev)
        nb = L.counter()
        ws = L.workspace(lib.pcc_rans_streams_ws_bytes(per, ns), dev)
        L.call("pcc_rans_encode_streams", L.ptr(sym), L.ptr(idx.contiguous()) if idx is not None else None, n, c, ng, segs,
               L.ptr(cdf), cdf.shape[1], L.ptr(sizes), L.ptr(offs), L.ptr(self._enc_table(dev)), L.ptr(out), L.cptr(nb),
               L.ptr(ws), ws.numel(), L.stream())
        from ...container import StreamBytes
        return StreamBytes(out[:int(L.read(nb)[0])].cpu().numpy().tobytes(), ng)

    # ---- the same coding, split so that a caller with several matrices to code shares the host reads --------------------
    HDR = 64                 # bytes in front of a container in its device blob: int64 nbytes | int64 guest word | padding
    FULL_COPY_BYTES = 1 << 20

    def streams_job(self, sym, idx=None):
        """Coding of one symbol matrix in steps (`StreamsJob`): estimate -> encode -> fetch, each step's device->host read
        exposed so that `MeanScaleHyperprior.compress` makes two reads for both strings instead of five."""
        return StreamsJob(self, sym, idx)

    # Speed / rate knob of the stream container.  A stream is one GPU lane running a serial recurrence (~0.2-0.3 us per
    # symbol) and costs 12 bytes of framing, so MORE streams is faster and FEWER is smaller.  The (n, c) rule above
    # keeps the framing small for a trained model's ~0.5 bpp frames; a frame that codes at a high rate (the benchmark's
    # random-weight frame: 8 bpp, 0.8 MB) can afford many more streams for the same RELATIVE overhead.  The encoder
    # therefore raises the segment count until the framing reaches FRAMING_TARGET of the payload, which it estimates from
    # the symbols themselves (`pcc_rans_estimate_bits`: one pass, integer sum -- the same symbols always give the same
    # stream count, so bitstreams stay reproducible).  Never below the (n, c) rule, never streams shorter than
    # MIN_SEGMENT_SYMBOLS; the count is written in the container, the decoder reads it there.
    FRAMING_TARGET = 0.02
    MIN_SEGMENT_SYMBOLS = 512
    ADAPTIVE_MIN_SYMBOLS = 1 << 18

    def _adaptive_segments(self, n, c, ng, segs, payload_bytes):
        per = n * (c // ng)
        want = int(payload_bytes * self.FRAMING_TARGET / 12) // ng
        return max(segs, min(want, max(1, self.MAX_STREAMS // ng), max(1, per // self.MIN_SEGMENT_SYMBOLS)))

    def _segments_of(self, data, n, c):
        """(groups, segments) of a container: the group count follows from (n, c), the segment count from the stream count
        in its first word (the encoder may have raised it, `_adaptive_segments`)."""
        ng, segs = self.n_streams(n, c)
        g = int(getattr(data, "groups", 0) or 0)           # a string read from a file carries its channel-group count
        if g > 0 and c % g == 0:
            ng = g
        if len(data) >= 4:
            ns = int.from_bytes(data[:4], "little")
            if ns >= ng and ns % ng == 0 and ns // ng <= max(1, self.MAX_STREAMS // ng):
                segs = ns // ng
        return ng, segs

    def decompress_rows(self, data, n, c, idx=None, device=None, check=None, status=None):
        """bytes -> sym [N,C] int32 on `device`.  `check`: list collecting the status words for a deferred check
        (keeps the decode path free of host synchronisation).  `status`: the counter() word to use (a caller that runs this
        under another stream allocates it beforehand: counter blocks are zeroed on the stream that is current when they are
        cut)."""
        self._check_tables()
        dev = torch.device(device) if device is not None else (idx.device if idx is not None else self._quantized_cdf.device)
        cdf, sizes, offs = (t.to(dev).contiguous() for t in (self._quantized_cdf, self._cdf_length, self._offset))
        if self.entropy_coder == "ans":
            i = (idx.t().contiguous().cpu().numpy().reshape(-1) if idx is not None
                 else np.repeat(np.arange(c, dtype=np.int32), n))
            s = self._host_decode(data, np.ascontiguousarray(i, np.int32))
            return torch.from_numpy(s.reshape(c, n).T.copy()).to(dev)
        buf = data.device_buf if isinstance(data, UploadedString) else self.upload_string(data, dev).device_buf
        sym = torch.empty((n, c), dtype=torch.int32, device=dev)
        if status is None:
            status = L.counter(1, torch.int32)
        L.call("pcc_rans_decode_streams", L.ptr(buf), len(data), L.ptr(idx.contiguous()) if idx is not None else None,
               n, c, *self._segments_of(data, n, c), L.ptr(cdf), cdf.shape[1], L.ptr(sizes), L.ptr(offs),
               L.ptr(self._dec_table(dev)), self._dec_table(dev).numel(), L.ptr(sym), L.ptr(status), L.stream())
        if check is None:                       # synchronous check (one device->host read)
            st = int(status.item())
            if st != 0:
                raise L.PccError(f"malformed rANS container (status {st})")
        else:                                   # deferred: the caller checks all status words once, at the end
            check.append(status)
        return sym

    @staticmethod
    def upload_string(data, dev):
        """The container's bytes on the device (padded: the decoder looks one word ahead), as an `UploadedString` that
        `decompress_rows` takes in place of the bytes -- so that a caller can start the copy early, on another stream."""
        if isinstance(data, UploadedString):
            return data
        nb = len(data)
        buf = torch.empty(nb + 8, dtype=torch.uint8, device=dev)
        if nb:
            with warnings.catch_warnings():          # (a read-only view of the bytes: no host-side copy before the upload)
                warnings.simplefilter("ignore")
                src = torch.frombuffer(data, dtype=torch.uint8)
            buf[:nb].copy_(src)
        buf[nb:].zero_()
        return UploadedString(data, buf)

    def _dec_table(self, dev):
        """Compact decoder table of the current CDFs (rebuilt when the tables change)."""
        tag = (self._tables_gen, self._quantized_cdf._version, str(dev))
        if getattr(self, "_dec_tag", None) != tag:
            cdf, sizes, _ = self._host_tables()
            lib = L.load()
            nb = lib.pcc_rans_dec_table_bytes(cdf.shape[0], _np_ptr(sizes))
            blob = np.zeros(nb, dtype=np.uint8)
            L.check(lib.pcc_rans_build_dec_table(_np_ptr(cdf), cdf.shape[0], cdf.shape[1], _np_ptr(sizes), _np_ptr(blob)),
                    "pcc_rans_build_dec_table")
            self._dec_dev = torch.from_numpy(blob).to(dev)
            self._dec_tag = tag
        return self._dec_dev

    def _enc_table(self, dev):
        """Division-free encoder entries of the current CDFs (rebuilt when the tables change)."""
        tag = (self._tables_gen, self._quantized_cdf._version, str(dev))
        if getattr(self, "_enc_tag", None) != tag:
            cdf, sizes, _ = self._host_tables()
            tab = np.zeros(cdf.shape[0] * cdf.shape[1] * 2, dtype=np.uint64)
            L.check(L.load().pcc_rans_build_enc_table(_np_ptr(cdf), cdf.shape[0], cdf.shape[1], _np_ptr(sizes),
                                                      _np_ptr(tab)), "pcc_rans_build_enc_table")
            self._enc_dev = torch.from_numpy(tab.view(np.int64)).to(dev)
            self._enc_tag = tag
        return self._enc_dev

    def _host_tables(self):
        return tuple(np.ascontiguousarray(t.detach().cpu().numpy().astype(np.int32))
                     for t in (self._quantized_cdf, self._cdf_length, self._offset))

    def _host_encode(self, sym, idx):
        cdf, sizes, offs = self._host_tables()
        lib = L.load()
        sym = np.ascontiguousarray(sym, np.int32)
        cap = lib.pcc_rans_max_bytes(len(sym))
        out = np.zeros(cap, np.uint8)
        nb = C.c_int64(0)
        L.check(lib.pcc_rans_encode_host(_np_ptr(sym), _np_ptr(idx), len(sym), _np_ptr(cdf), cdf.shape[1], _np_ptr(sizes),
                                         _np_ptr(offs), _np_ptr(out), cap, C.byref(nb)), "pcc_rans_encode_host")
        return out[:nb.value].tobytes()

    def _host_decode(self, data, idx):
        cdf, sizes, offs = self._host_tables()
        buf = np.frombuffer(data, np.uint8).copy()
        out = np.zeros(len(idx), np.int32)
        L.check(L.load().pcc_rans_decode_host(_np_ptr(buf), len(buf), _np_ptr(idx), len(idx), _np_ptr(cdf), cdf.shape[1],
                                              _np_ptr(sizes), _np_ptr(offs), _np_ptr(out)), "pcc_rans_decode_host")
        return out

    # ---- CompressAI surface: [B, C, N] tensors, one string per batch element ----------------------------------
    def compress(self, inputs, indexes, means=None):
        """`EntropyModel.compress` (`model/entropy_models.py:397-400`): symbols = round(inputs - means), one string
        per batch element over the channel-major flattening."""
        sym = torch.round(inputs - means if means is not None else inputs).to(torch.int32)
        out = []
        for b in range(sym.shape[0]):
            s = sym[b].reshape(sym.shape[1], -1).t().contiguous()               # [N, C]
            i = indexes[b].reshape(sym.shape[1], -1).t().contiguous().to(torch.int32)
            out.append(self.compress_rows(s, i))
        return out

    def decompress(self, strings, indexes, dtype=torch.float, means=None):
        """`EntropyModel.decompress` (`model/entropy_models.py:471,484`)."""
        outs = []
        for b, data in enumerate(strings):
            i = indexes[b].reshape(indexes.shape[1], -1).t().contiguous().to(torch.int32)
            s = self.decompress_rows(data, i.shape[0], i.shape[1], i)
            outs.append(s.t().reshape(indexes.shape[1:]))
        out = torch.stack(outs, dim=0).to(dtype)
        return out + means if means is not None else out


class UploadedString:
    """A coded string together with its (padded) copy in device memory; behaves like the bytes for the host-side parsing."""

    def __init__(self, data, device_buf):
        self.data, self.device_buf = data, device_buf
        self.groups = getattr(data, "groups", 0)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, i):
        return self.data[i]


class StreamsJob:
    """GPU rANS coding of one [N, C] symbol matrix (`EntropyModel.compress_rows`, entropy_coder "pcc_streams") in steps.

      adaptive            the stream count depends on a payload estimate (`_adaptive_segments`)
      launch_estimate(p)  queue the estimate kernel; its int64 result goes to device address p (zeroed by the caller)
      launch_encode(est)  queue the encoder into this job's blob = [HDR bytes | container]; word 0 of the header is the
                          container's byte count, word 1 is free for a guest (another job's estimate)
      fetch()             one device->host copy: header + container (large containers: header + the estimated size, a second
                          copy only if the estimate fell short); returns the bytes
    """

    def __init__(self, em, sym, idx):
        em._check_tables()
        self.em, self.sym = em, sym.contiguous()
        self.idx = idx.contiguous() if idx is not None else None
        self.n, self.c = self.sym.shape
        dev = self.sym.device
        self.tabs = tuple(t.to(dev).contiguous() for t in (em._quantized_cdf, em._cdf_length, em._offset))
        self.ng, self.segs = em.n_streams(self.n, self.c)
        self.adaptive = em.FRAMING_TARGET > 0 and self.n * self.c >= em.ADAPTIVE_MIN_SYMBOLS
        self.blob, self.guess, self.host = None, None, None

    def launch_estimate(self, d_ptr):
        cdf, sizes, offs = self.tabs
        L.call("pcc_rans_estimate_bits", L.ptr(self.sym), L.ptr(self.idx), self.n, self.c, L.ptr(cdf), cdf.shape[1], L.ptr(sizes),
               L.ptr(offs), d_ptr, L.stream())

    def launch_encode(self, est_bits256=None):
        em, dev = self.em, self.sym.device
        lib = L.load()
        cdf, sizes, offs = self.tabs
        if self.adaptive:
            if est_bits256 is None:
                raise L.PccError("StreamsJob: an adaptive container needs its payload estimate first")
            payload = int(est_bits256) / 2048.0
            self.segs = em._adaptive_segments(self.n, self.c, self.ng, self.segs, payload)
        ns = self.ng * self.segs
        per = lib.pcc_rans_stream_symbols(self.n, self.c, self.ng, self.segs)
        cap = lib.pcc_rans_container_max_bytes(per, ns)
        self.blob = torch.empty(em.HDR + cap, dtype=torch.uint8, device=dev)
        self.blob[:em.HDR].zero_()
        if self.adaptive and cap > em.FULL_COPY_BYTES:       # payload estimate + framing + slack: what the first copy fetches
            self.guess = min(cap, int(payload * 1.02) + 12 * ns + 4096)
        ws = L.workspace(lib.pcc_rans_streams_ws_bytes(per, ns), dev)
        L.call("pcc_rans_encode_streams", L.ptr(self.sym), L.ptr(self.idx), self.n, self.c, self.ng, self.segs, L.ptr(cdf),
               cdf.shape[1], L.ptr(sizes), L.ptr(offs), L.ptr(em._enc_table(dev)), self.blob.data_ptr() + em.HDR,
               self.blob.data_ptr(), L.ptr(ws), ws.numel(), L.stream())
        return self

    def guest_ptr(self):
        return self.blob.data_ptr() + 8

    def attach(self, word):
        """Carry a 4-byte device word (int32 [1]) home in header bytes 16-19 of the next fetch (`self.guest2`)."""
        self.blob[16:20].copy_(word.view(torch.uint8))

    def fetch(self):
        H = self.em.HDR
        take = self.blob.numel() if self.guess is None else H + self.guess
        host = self.blob[:take].cpu().numpy()
        nbytes = int(host[:8].view(np.int64)[0])
        self.guest = int(host[8:16].view(np.int64)[0])
        self.guest2 = int(host[16:20].view(np.int32)[0])
        if nbytes > take - H:                                   # the estimate fell short: fetch the rest
            host = self.blob[:H + nbytes].cpu().numpy()
        from ...container import StreamBytes
        return StreamBytes(host[H:H + nbytes].tobytes(), self.ng)


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

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        """Accept both spellings of the factorised-prior parameters: `_matrix{i}` / `_bias{i}` / `_factor{i}` (CompressAI
        <= 1.2.0 attribute names, used here) and `matrices.{i}` / `biases.{i}` / `factors.{i}` (the ParameterList names
        SURVEY B.4 records for 1.2.4), so a checkpoint written by either loads."""
        for i in range(len(self.filters) + 1):
            for new, old in ((f"matrices.{i}", f"_matrix{i}"), (f"biases.{i}", f"_bias{i}"), (f"factors.{i}", f"_factor{i}")):
                if prefix + new in state_dict and prefix + old not in state_dict:
                    state_dict[prefix + old] = state_dict.pop(prefix + new)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def _get_medians(self):
        return self.quantiles[:, :, 1:2].detach()

    def _logits_cumulative(self, inputs, stop_gradient=False):
        logits = inputs
        for i in range(len(self.filters) + 1):
            m = getattr(self, f"_matrix{i:d}")
            b = getattr(self, f"_bias{i:d}")
            if stop_gradient:
                m, b = m.detach(), b.detach()
            sp = torch.nn.functional.softplus(m)
            if logits.is_cuda and logits.shape[-1] <= 8:
                # the quantile loss' [C, f_out, f_in] x [C, f_in, 3] products: written as a broadcast product and a sum over the
                # (<= 3) inner terms -- a batched GEMM of this size costs the host ~40 us per call in the BLAS dispatch, ten calls
                # per training step, with the GPU idle behind each
                logits = (sp.unsqueeze(-1) * logits.unsqueeze(1)).sum(dim=2) + b
            else:
                logits = torch.matmul(sp, logits) + b
            if i < len(self.filters):
                fa = getattr(self, f"_factor{i:d}")
                if stop_gradient:
                    fa = fa.detach()
                logits = logits + torch.tanh(fa) * torch.tanh(logits)
        return logits

    def loss(self):
        logits = self._logits_cumulative(self.quantiles, stop_gradient=True)
        return torch.abs(logits - self.target).sum()

    @torch.no_grad()
    def update(self, force=False):
        """`EntropyBottleneck.update` (CompressAI): per channel the pmf over [median - minima, median + maxima] and its
        quantised CDF (`model/model.py:30-34` calls this through `CompressionModel.update`).  Evaluated on the host
        once per parameter update."""
        if self._offset.numel() > 0 and not force:
            return False
        dev = self.quantiles.device
        cpu = {k: v.detach().cpu().float() for k, v in self.named_parameters()}
        q = cpu["quantiles"]
        med = q[:, 0, 1]
        minima = torch.ceil(med - q[:, 0, 0]).int().clamp(min=0)
        maxima = torch.ceil(q[:, 0, 2] - med).int().clamp(min=0)
        pmf_start = med - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max().item())
        samples = (torch.arange(max_length)[None, :] + pmf_start[:, None])[:, None, :]

        def logits(x):
            for i in range(len(self.filters) + 1):
                x = torch.matmul(torch.nn.functional.softplus(cpu[f"_matrix{i}"]), x) + cpu[f"_bias{i}"]
                if i < len(self.filters):
                    x = x + torch.tanh(cpu[f"_factor{i}"]) * torch.tanh(x)
            return x
        lower, upper = logits(samples - 0.5), logits(samples + 0.5)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail, pmf_length, max_length).to(dev)
        self._offset = (-minima).int().to(dev)
        self._cdf_length = (pmf_length + 2).int().to(dev)
        self._tables_gen += 1
        return True

    def _packed_with_grad(self):
        """`packed()` under autograd: [C, 58] = softplus(matrices) | biases | tanh(factors) (14 small ops; their derivatives
        are torch's, the likelihood's own gradient comes from `pcc_eb_lik_bwd`)."""
        parts = [torch.nn.functional.softplus(getattr(self, f"_matrix{i}")).reshape(self.channels, -1) for i in range(5)]
        parts += [getattr(self, f"_bias{i}").reshape(self.channels, -1) for i in range(5)]
        parts += [torch.tanh(getattr(self, f"_factor{i}")).reshape(self.channels, -1) for i in range(4)]
        return torch.cat(parts, dim=1).to(torch.float32)

    def likelihood_rows(self, v):
        """Differentiable likelihood of [N,C] rows (training path): |sigmoid(s*u) - sigmoid(s*l)| >= 1e-9 -- one HIP kernel per
        direction on the GPU (`pcc_eb_lik_fwd/bwd`; filters (3,3,3,3), bound 1e-9 as the reference constructs it), the torch
        chain otherwise."""
        if (v.is_cuda and v.dtype == torch.float32 and self.filters == (3, 3, 3, 3) and self.use_likelihood_bound
                and abs(self._likelihood_bound_value - 1e-9) < 1e-15 and v.dim() == 2 and v.shape[1] == self.channels):
            from ...autograd import EbLikFn
            return EbLikFn.apply(v, self._packed_with_grad())
        x = v.t().unsqueeze(1)
        lo, up = self._logits_cumulative(x - 0.5), self._logits_cumulative(x + 0.5)
        sg = -torch.sign(lo + up).detach()
        lik = torch.abs(torch.sigmoid(sg * up) - torch.sigmoid(sg * lo))[:, 0, :].t()
        return self.likelihood_lower_bound(lik) if self.use_likelihood_bound else lik

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

    def compress(self, x):
        """`EntropyBottleneck.compress` (`model/entropy_models.py:371`): x [B,C,N] -> one string per batch element."""
        out = []
        for b in range(x.shape[0]):
            sym, _, _ = self.encode_rows(x[b].reshape(x.shape[1], -1).t().contiguous(), want_likelihood=False)
            out.append(self.compress_rows(sym))
        return out

    def decompress(self, strings, size):
        """`EntropyBottleneck.decompress` (`model/entropy_models.py:372,438`): -> [B, C, *size] float."""
        n = int(np.prod(size))
        med = self.quantiles[:, 0, 1].detach().to(torch.float32)
        outs = [self.decompress_rows(s, n, self.channels, device=med.device).to(torch.float32) + med[None, :]
                for s in strings]
        return torch.stack([o.t().reshape(self.channels, *size) for o in outs], dim=0)

    noise_fn = None     # callable(like) -> U(-.5,.5) noise; tests install a deterministic one

    def forward(self, x, training=None):
        """`EntropyBottleneck.forward` ([B,C,N]; `model/entropy_models.py:272,282`): training -> additive uniform noise
        (`quantize(.., "noise")`), eval -> round(x - median) + median; likelihood of the result, floored at 1e-9."""
        training = self.training if training is None else training
        rows, back = _rows(x)
        if training:
            noise = self.noise_fn(rows) if self.noise_fn is not None else torch.empty_like(rows).uniform_(-0.5, 0.5)
            out = rows + noise
            return back(out), back(self.likelihood_rows(out))
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
        self._scale_bound_value = float(scale_bound) if scale_bound is not None else None
        self.register_buffer("scale_table", torch.Tensor(tuple(float(s) for s in scale_table)) if scale_table else torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]) if scale_bound is not None else None)

    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        dev = self.scale_table.device
        self.scale_table = torch.as_tensor(scale_table, dtype=torch.float32).to(dev)
        self.update()
        return True

    @staticmethod
    def _standardized_cumulative(x):
        return 0.5 * torch.erfc(-(2 ** -0.5) * x)

    @torch.no_grad()
    def update(self):
        """`GaussianConditional.update` (CompressAI): per table scale the pmf over [-c, c], c = ceil(scale * 6.11), and
        its quantised CDF with the tail mass as bypass sentinel."""
        from scipy.stats import norm
        dev = self.scale_table.device
        st = self.scale_table.detach().cpu().float()
        multiplier = -norm.ppf(self.tail_mass / 2)
        pmf_center = torch.ceil(st * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = int(pmf_length.max().item())
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        scale = st[:, None]
        upper = self._standardized_cumulative((0.5 - samples) / scale)
        lower = self._standardized_cumulative((-0.5 - samples) / scale)
        pmf = upper - lower
        tail = 2 * lower[:, :1]
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail, pmf_length, max_length).to(dev)
        self._offset = (-pmf_center).int().to(dev)
        self._cdf_length = (pmf_length + 2).int().to(dev)
        self._tables_gen += 1

    def _table(self, device):
        if self.scale_table.numel() == 0:
            raise L.PccError("GaussianConditional has no scale table: call model.update() first (`evaluate.py:89`)")
        return self.scale_table.to(device=device, dtype=torch.float32).contiguous()

    def likelihood_rows(self, values, scales, means):
        """Differentiable Gaussian likelihood of [N,C] rows (training path): one HIP kernel per direction on the GPU
        (`pcc_gauss_lik_fwd/bwd`, bounds 0.11 / 1e-9 as constructed by the reference); torch ops otherwise."""
        if (values.is_cuda and self.use_likelihood_bound and values.dtype == torch.float32
                and abs(self._likelihood_bound_value - 1e-9) < 1e-15                         # (the kernels hard-code both bounds: 0.11 / 1e-9)
                and self._scale_bound_value is not None and abs(self._scale_bound_value - 0.11) < 1e-7 and values.shape == scales.shape == means.shape):
            from ...autograd import GaussLikFn
            return GaussLikFn.apply(values, scales, means)
        s = self.lower_bound_scale(scales)
        a = torch.abs(values - means)
        lik = self._standardized_cumulative((0.5 - a) / s) - self._standardized_cumulative((-0.5 - a) / s)
        return self.likelihood_lower_bound(lik) if self.use_likelihood_bound else lik

    def index_rows(self, params, keys=None, gain=None):
        """Table rows of every element from (scales_hat | means_hat) alone: what the decoder needs before the symbols
        exist (`model/entropy_models.py:468`)."""
        return self.encode_rows(None, params, keys, gain, want_likelihood=False, want_symbols=False)[1]

    def encode_rows(self, y, params, keys=None, gain=None, want_likelihood=True, want_symbols=True):
        """Fused a7 kernel on [N,C] rows; params [N,2C] = (scales_hat | means_hat)."""
        params = params.contiguous()
        n, c = params.shape[0], params.shape[1] // 2
        if y is None:
            assert not want_likelihood and not want_symbols
        else:
            y = y.contiguous()
        tab = self._table(params.device)
        sym = torch.empty((n, c), dtype=torch.int32, device=params.device) if want_symbols else None
        idx = torch.empty((n, c), dtype=torch.int32, device=params.device)
        lik = torch.empty((n, c), dtype=torch.float32, device=params.device) if want_likelihood else None
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
        """Eval: (round(x - mu) + mu, likelihood); training: (x + U(-.5,.5), likelihood)
        (`model/entropy_models.py:312-316,329-333`)."""
        training = self.training if training is None else training
        x, back = _rows(inputs)
        s, _ = _rows(scales)
        m = _rows(means)[0] if means is not None else torch.zeros_like(x)
        if training:          # `quantize(inputs, "noise", means)`: inputs + U(-.5,.5) (`model/entropy_models.py:312-316,327-331`)
            noise = self.noise_fn(x) if self.noise_fn is not None else torch.empty_like(x).uniform_(-0.5, 0.5)
            out = x + noise
            return back(out), back(self.likelihood_rows(out, s, m))
        sym, _, lik = self.encode_rows(x, torch.cat([s, m], dim=1))
        return back(sym.to(torch.float32) + m), back(lik)

    noise_fn = None
