"""Entropy-model arithmetic of the oracle (numpy fp32).  Test infrastructure.

Restates CompressAI 1.2.4 (`requirements.txt:9`, not vendored in the reference)
as used by `model/entropy_models.py`; formulas per SURVEY.md Appendix B.3/B.4.
"""
import numpy as np
from scipy.special import erfc, expit

F32 = np.float32
SCALE_MIN, SCALE_MAX, SCALE_LEVELS = 0.11, 256.0, 64
SCALE_BOUND = 0.11
LIKELIHOOD_BOUND = 1e-9


def scale_table():
    """CompressAI `get_scale_table()` installed by `CompressionModel.update`
    (`model/model.py:30-34`): exp(linspace(ln .11, ln 256, 64))."""
    return np.exp(np.linspace(np.log(SCALE_MIN), np.log(SCALE_MAX), SCALE_LEVELS)).astype(F32)


def lower_bound(x, b):
    """`compressai.ops.LowerBound` forward."""
    return np.maximum(np.asarray(x, dtype=F32), F32(b))


def build_indexes(scales, table=None):
    """`GaussianConditional.build_indexes` (`model/entropy_models.py:396,468`):
    idx = 63 - #{t < 63 : max(scale,0.11) <= table[t]}."""
    table = scale_table() if table is None else np.asarray(table, dtype=F32)
    s = lower_bound(scales, SCALE_BOUND)
    idx = np.full(s.shape, len(table) - 1, dtype=np.int32)
    for t in table[:-1]:
        idx -= (s <= t).astype(np.int32)
    return idx


def std_cumulative(x):
    """`GaussianConditional._standardized_cumulative`: 0.5*erfc(-x/sqrt(2))."""
    x = np.asarray(x, dtype=F32)
    return (F32(0.5) * erfc(F32(-(2 ** -0.5)) * x)).astype(F32)


def gaussian_likelihood(values, scales, means=None):
    """`GaussianConditional._likelihood` + likelihood lower bound 1e-9
    (`model/entropy_models.py:299-333`): Phi((.5-|v|)/s) - Phi((-.5-|v|)/s)."""
    v = np.asarray(values, dtype=F32)
    if means is not None:
        v = v - np.asarray(means, dtype=F32)
    s = lower_bound(scales, SCALE_BOUND)
    v = np.abs(v)
    upper = std_cumulative((F32(0.5) - v) / s)
    lower = std_cumulative((F32(-0.5) - v) / s)
    return np.maximum(upper - lower, F32(LIKELIHOOD_BOUND)).astype(F32)


def quantize_symbols(x, means=None):
    """`EntropyModel.quantize(mode="symbols")`: round-half-even(x - means) as int32
    (`model/entropy_models.py:397-400` via `gaussian_conditional.compress`)."""
    x = np.asarray(x, dtype=F32)
    if means is not None:
        x = x - np.asarray(means, dtype=F32)
    return np.rint(x).astype(np.int32)


def dequantize(symbols, means=None):
    """`EntropyModel.dequantize`: symbols.float() + means (`model/entropy_models.py:484`)."""
    out = np.asarray(symbols).astype(F32)
    if means is not None:
        out = out + np.asarray(means, dtype=F32)
    return out


# --------------------------------------------------------------------------
# Factorised prior (EntropyBottleneck), SURVEY B.4
# --------------------------------------------------------------------------

def eb_init(channels, seed=0, init_scale=10.0, filters=(3, 3, 3, 3)):
    """Parameter set with CompressAI's shapes/initial values (bias ~ U(-.5,.5))."""
    rng = np.random.default_rng(seed)
    f = (1,) + tuple(filters) + (1,)
    scale = init_scale ** (1.0 / (len(filters) + 1))
    p = {}
    for i in range(len(filters) + 1):
        init = np.log(np.expm1(1.0 / scale / f[i + 1]))
        p[f"_matrix{i}"] = np.full((channels, f[i + 1], f[i]), init, dtype=F32)
        p[f"_bias{i}"] = rng.uniform(-0.5, 0.5, (channels, f[i + 1], 1)).astype(F32)
        if i < len(filters):
            p[f"_factor{i}"] = np.zeros((channels, f[i + 1], 1), dtype=F32)
    q = np.tile(np.array([-init_scale, 0.0, init_scale], dtype=F32), (channels, 1, 1))
    p["quantiles"] = q
    return p


def _softplus(x):
    x = np.asarray(x, dtype=F32)
    return (np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))).astype(F32)


def eb_logits_cumulative(p, x):
    """`EntropyBottleneck._logits_cumulative`; x: [C,1,N] float32."""
    logits = np.asarray(x, dtype=F32)
    n = sum(1 for k in p if k.startswith("_matrix"))
    for i in range(n):
        m = _softplus(p[f"_matrix{i}"])
        logits = np.matmul(m, logits) + p[f"_bias{i}"]
        if i < n - 1:
            logits = logits + np.tanh(p[f"_factor{i}"]) * np.tanh(logits)
        logits = logits.astype(F32)
    return logits


def eb_medians(p):
    """`EntropyBottleneck._get_medians` (`model/entropy_models.py:283`): quantiles[:,:,1:2]."""
    return p["quantiles"][:, :, 1:2]


def eb_likelihood(p, values):
    """`EntropyBottleneck._likelihood` + 1e-9 bound; values: [C,N] (already quantised)."""
    v = np.asarray(values, dtype=F32)[:, None, :]
    lower = eb_logits_cumulative(p, v - F32(0.5))
    upper = eb_logits_cumulative(p, v + F32(0.5))
    sign = -np.sign(lower + upper)
    lik = np.abs(expit(sign * upper) - expit(sign * lower)).astype(F32)
    return np.maximum(lik[:, 0, :], F32(LIKELIHOOD_BOUND))


def eb_quantize(p, z):
    """Eval-mode quantisation of the hyper latent (`model/entropy_models.py:371-372`):
    symbols = round(z - median), z_hat = symbols + median.  z: [C,N]."""
    med = eb_medians(p)[:, 0, :]
    sym = np.rint(np.asarray(z, dtype=F32) - med).astype(np.int32)
    return sym, (sym.astype(F32) + med).astype(F32)
