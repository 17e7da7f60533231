"""rANS entropy coder + quantised CDF tables of the oracle (pure Python / numpy).  Test infrastructure.

Restates what the reference obtains from CompressAI 1.2.4's C++ extension (`compressai/cpp_exts/rans/rans_interface.cpp`
on ryg_rans `rans64.h`, and `pmf_to_quantized_cdf` in `cpp_exts/ops/ops.cpp`; neither is vendored in the reference) at
the call sites `model/entropy_models.py:371-372,397-400,438,471,484` and `model/model.py:30-34`:
64-bit state, 32-bit renormalisation words, 16-bit probability precision, 4-bit bypass digits for out-of-table
values, symbols consumed in reverse so the decoder reads forward.  PARITY UNPINNED against CompressAI's bytes (the
extension cannot run here); round-trip and cross-implementation (host C++ == GPU == this file) equality are tested.
"""
import numpy as np
from scipy.stats import norm

PRECISION = 16
BYPASS_PRECISION = 4
MAX_BYPASS = (1 << BYPASS_PRECISION) - 1
RANS_L = 1 << 31
MASK32 = 0xFFFFFFFF


def pmf_to_quantized_cdf(pmf, precision=PRECISION):
    """`pmf_to_quantized_cdf` (ops.cpp): round to `precision` bits, renormalise, make every frequency >= 1 by
    stealing from the smallest frequency > 1."""
    pmf = np.asarray(pmf, dtype=np.float32)
    cdf = np.zeros(len(pmf) + 1, dtype=np.int64)
    cdf[1:] = np.round(pmf.astype(np.float64) * (1 << precision)).astype(np.int64)   # std::round of float * int
    total = int(cdf.sum())
    assert total > 0
    cdf = ((1 << precision) * cdf) // total
    cdf = np.cumsum(cdf)
    cdf[-1] = 1 << precision
    n = len(cdf)
    for i in range(n - 1):
        if cdf[i] == cdf[i + 1]:
            best_freq, best = None, -1
            for j in range(n - 1):
                f = cdf[j + 1] - cdf[j]
                if f > 1 and (best_freq is None or f < best_freq):
                    best_freq, best = f, j
            assert best != -1
            if best < i:
                cdf[best + 1:i + 1] -= 1
            else:
                cdf[i + 1:best + 1] += 1
    return cdf.astype(np.int32)


def gaussian_tables(scale_table, tail_mass=1e-9):
    """`GaussianConditional.update()` (CompressAI): per scale a pmf over [-c, c], c = ceil(scale * -ppf(tail/2)),
    plus the tail mass as the bypass sentinel.  Returns (cdf [S, L+2] int32, cdf_length [S], offset [S])."""
    from .entropy import std_cumulative
    st = np.asarray(scale_table, dtype=np.float32)
    multiplier = np.float32(-norm.ppf(tail_mass / 2))
    center = np.ceil(st * multiplier).astype(np.int32)
    length = 2 * center + 1
    max_len = int(length.max())
    samples = np.abs(np.arange(max_len, dtype=np.int32)[None, :] - center[:, None]).astype(np.float32)
    s = st[:, None]
    upper = std_cumulative((np.float32(0.5) - samples) / s)
    lower = std_cumulative((np.float32(-0.5) - samples) / s)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    cdf = np.zeros((len(st), max_len + 2), dtype=np.int32)
    for i in range(len(st)):
        prob = np.concatenate([pmf[i, :length[i]], tail[i]])
        c = pmf_to_quantized_cdf(prob)
        cdf[i, :len(c)] = c
    return cdf, (length + 2).astype(np.int32), (-center).astype(np.int32)


def bottleneck_tables(p):
    """`EntropyBottleneck.update()` (CompressAI): per channel pmf over [median - minima, median + maxima]."""
    from . import entropy as en
    from scipy.special import expit
    q = np.asarray(p["quantiles"], dtype=np.float32)
    med = q[:, 0, 1]
    minima = np.maximum(np.ceil(med - q[:, 0, 0]).astype(np.int32), 0)
    maxima = np.maximum(np.ceil(q[:, 0, 2] - med).astype(np.int32), 0)
    pmf_start = med - minima
    length = maxima + minima + 1
    max_len = int(length.max())
    samples = (np.arange(max_len, dtype=np.float32)[None, :] + pmf_start[:, None]).astype(np.float32)[:, None, :]
    lower = en.eb_logits_cumulative(p, samples - np.float32(0.5))
    upper = en.eb_logits_cumulative(p, samples + np.float32(0.5))
    sign = -np.sign(lower + upper)
    pmf = np.abs(expit(sign * upper) - expit(sign * lower)).astype(np.float32)[:, 0, :]
    tail = (expit(lower[:, 0, :1]) + expit(-upper[:, 0, -1:])).astype(np.float32)
    cdf = np.zeros((len(med), max_len + 2), dtype=np.int32)
    for i in range(len(med)):
        prob = np.concatenate([pmf[i, :length[i]], tail[i]])
        c = pmf_to_quantized_cdf(prob)
        cdf[i, :len(c)] = c
    return cdf, (length + 2).astype(np.int32), (-minima).astype(np.int32)


def _sub_symbols(sym, idx, cdf, sizes, offsets):
    """Forward list of (start, range, bypass) entries `encode_with_indexes` pushes for one symbol."""
    c = cdf[idx]
    max_value = int(sizes[idx]) - 2
    value = int(sym) - int(offsets[idx])
    raw = 0
    if value < 0:
        raw = -2 * value - 1
        value = max_value
    elif value >= max_value:
        raw = 2 * (value - max_value)
        value = max_value
    out = [(int(c[value]), int(c[value + 1]) - int(c[value]), False)]
    if value == max_value:
        nb = 0
        while (raw >> (nb * BYPASS_PRECISION)) != 0:
            nb += 1
        val = nb
        while val >= MAX_BYPASS:
            out.append((MAX_BYPASS, MAX_BYPASS + 1, True))
            val -= MAX_BYPASS
        out.append((val, val + 1, True))
        for j in range(nb):
            v = (raw >> (j * BYPASS_PRECISION)) & MAX_BYPASS
            out.append((v, v + 1, True))
    return out


def encode(symbols, indexes, cdf, sizes, offsets):
    """`BufferedRansEncoder.encode_with_indexes` + `flush` -> bytes (little-endian 32-bit words)."""
    subs = []
    for s, i in zip(np.asarray(symbols).tolist(), np.asarray(indexes).tolist()):
        subs.extend(_sub_symbols(s, i, cdf, sizes, offsets))
    x = RANS_L
    words = []
    for start, rng, bypass in reversed(subs):
        if not bypass:
            x_max = ((RANS_L >> PRECISION) << 32) * rng
            if x >= x_max:
                words.append(x & MASK32)
                x >>= 32
            x = ((x // rng) << PRECISION) + (x % rng) + start
        else:
            freq = 1 << (16 - BYPASS_PRECISION)
            x_max = ((RANS_L >> 16) << 32) * freq
            if x >= x_max:
                words.append(x & MASK32)
                x >>= 32
            x = (x << BYPASS_PRECISION) | start
    words.append((x >> 32) & MASK32)
    words.append(x & MASK32)
    return np.array(words[::-1], dtype="<u4").tobytes()


def decode(data, indexes, cdf, sizes, offsets):
    """`RansDecoder.decode_with_indexes` -> int32 symbols."""
    w = np.frombuffer(data, dtype="<u4").astype(np.uint64).tolist()
    x = int(w[0]) | (int(w[1]) << 32)
    p = 2
    out = []

    def get_bits():
        nonlocal x, p
        v = x & MAX_BYPASS
        x >>= BYPASS_PRECISION
        if x < RANS_L:
            x = (x << 32) | int(w[p])
            p += 1
        return v

    for i in np.asarray(indexes).tolist():
        c = cdf[i]
        size = int(sizes[i])
        max_value = size - 2
        cum = x & ((1 << PRECISION) - 1)
        s = int(np.searchsorted(c[:size], cum, side="right")) - 1
        start, rng = int(c[s]), int(c[s + 1]) - int(c[s])
        x = rng * (x >> PRECISION) + cum - start
        if x < RANS_L:
            x = (x << 32) | int(w[p])
            p += 1
        value = s
        if value == max_value:
            val = get_bits()
            nb = val
            while val == MAX_BYPASS:
                val = get_bits()
                nb += val
            raw = 0
            for j in range(nb):
                raw |= get_bits() << (j * BYPASS_PRECISION)
            value = raw >> 1
            value = -value - 1 if (raw & 1) else value + max_value
        out.append(value + int(offsets[i]))
    return np.array(out, dtype=np.int32)
