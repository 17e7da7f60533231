"""End-to-end restatement of the reference codec flow (numpy).  Test infrastructure.

Follows, line by line in behaviour (not in code), the reference's
`AnalysisTransform.forward` (`model/transforms.py:68-97`),
`SparseSynthesisTransform.forward` (`model/transforms.py:170-225`),
`MeanScaleHyperprior.compress/decompress` (`model/entropy_models.py:344-490`) and
`UnifiedModel.compress/decompress` (`model/model.py:94-250`) on top of the
operator semantics in `oracle.coords` / `oracle.ops` / `oracle.entropy`.

Parameters are a flat dict keyed by the reference's `state_dict` names
(e.g. ``g_a.down_conv_1.0.kernel``), values numpy float32.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import coords as co
from . import entropy as en
from . import ops

F32 = np.float32


# --------------------------------------------------------------------------
# kernel maps in pair form (memory-lean for the 125-offset generative layers)
# --------------------------------------------------------------------------

def kernel_map_pairs(in_keys, out_keys, kernel_size, step, transposed=False, threads=1):
    """Per kernel offset k: (in_rows, out_rows) int32 arrays, out_rows ascending.
    Same relation as `coords.kernel_map`."""
    d = co.offset_deltas(co.kernel_offsets(kernel_size), step)
    in_keys = np.asarray(in_keys, dtype=np.int64)
    out_keys = np.asarray(out_keys, dtype=np.int64)

    def one(k):
        if transposed:
            # every (in,k) is exactly one pair: out = in + off_k*step
            o = co.lookup(out_keys, in_keys + d[k])
            i = np.nonzero(o >= 0)[0].astype(np.int32)
            o = o[i]
            order = np.argsort(o, kind="stable")
            return i[order], o[order]
        i = co.lookup(in_keys, out_keys + d[k])
        o = np.nonzero(i >= 0)[0].astype(np.int32)
        return i[o], o

    if threads > 1:
        with ThreadPoolExecutor(threads) as ex:
            return list(ex.map(one, range(d.shape[0])))
    return [one(k) for k in range(d.shape[0])]


def pairs_to_nbr(pairs, n_out):
    nbr = np.full((len(pairs), n_out), -1, dtype=np.int32)
    for k, (i, o) in enumerate(pairs):
        nbr[k, o] = i
    return nbr


def conv_pairs(feats, W, bias, pairs, n_out):
    """`ops.conv` on pair-form maps: out[o] = bias + sum_k feats[i] @ W[k]."""
    feats = np.ascontiguousarray(feats, dtype=F32)
    W = np.asarray(W, dtype=F32)
    if W.ndim == 2:
        W = W[None]
    out = np.zeros((n_out, W.shape[2]), dtype=F32)
    if bias is not None:
        out += np.asarray(bias, dtype=F32).reshape(1, -1)
    for k, (i, o) in enumerate(pairs):
        if i.size:
            out[o] += feats[i] @ W[k]
    return out


def _conv_layer(P, name, keys, feats, ts, kernel_size, stride=1, transposed=False,
                bias=True, stats=None, threads=1):
    """Apply conv `name` (state-dict prefix) to (keys, feats, tensor_stride)."""
    W = P[name + ".kernel"]
    b = P.get(name + ".bias") if bias else None
    if transposed:
        ts_out = ts // stride
        out_keys = co.expand_keys(keys, kernel_size, ts_out)
        pairs = kernel_map_pairs(keys, out_keys, kernel_size, ts_out, transposed=True, threads=threads)
    else:
        ts_out = ts * stride
        out_keys = keys if stride == 1 else co.stride_keys(keys, ts_out)
        if kernel_size == 1:
            ar = np.arange(len(keys), dtype=np.int32)
            pairs = [(ar, ar)]
        else:
            pairs = kernel_map_pairs(keys, out_keys, kernel_size, ts, threads=threads)
    out = conv_pairs(feats, W, b, pairs, len(out_keys))
    if stats is not None:
        Wd = W if W.ndim == 3 else W[None]
        stats.append(dict(name=name, n_in=len(keys), n_out=len(out_keys), K=Wd.shape[0],
                          pairs=int(sum(len(i) for i, _ in pairs)), cin=Wd.shape[1], cout=Wd.shape[2]))
    return out_keys, out, ts_out


def _gdn(P, name, feats, inverse):
    return ops.gdn(feats, P[name + ".beta"], P[name + ".gamma"], inverse=inverse)


def _batch_of(keys):
    return (np.asarray(keys, dtype=np.int64) >> 48).astype(np.int64)


def count_per_batch(keys):
    """`AnalysisTransform.count_per_batch` (`model/transforms.py:47-64`)."""
    b = _batch_of(keys)
    return [int((b == v).sum()) for v in np.unique(b)]


# --------------------------------------------------------------------------
# transforms
# --------------------------------------------------------------------------

def analysis(P, keys, feats, stats=None, threads=1):
    """g_a (`model/transforms.py:68-97`).  Returns y_keys, y_feats, k."""
    k = [count_per_batch(keys)]
    ts = 1
    keys, f, ts = _conv_layer(P, "g_a.down_conv_1.0", keys, feats, ts, 5, 2, stats=stats, threads=threads)
    f = _gdn(P, "g_a.down_conv_1.1", f, False)
    k.append(count_per_batch(keys))
    keys, f, ts = _conv_layer(P, "g_a.down_conv_2.0", keys, f, ts, 5, 2, stats=stats, threads=threads)
    f = _gdn(P, "g_a.down_conv_2.1", f, False)
    k.append(count_per_batch(keys))
    keys, f, ts = _conv_layer(P, "g_a.down_conv_3.0", keys, f, ts, 5, 2, stats=stats, threads=threads)
    f = _gdn(P, "g_a.down_conv_3.1", f, False)
    keys, f, ts = _conv_layer(P, "g_a.down_conv_3.2", keys, f, ts, 5, 1, stats=stats, threads=threads)
    k.reverse()
    return keys, f, k


def _predict(P, name, keys, f, ts, stats, threads):
    _, h, _ = _conv_layer(P, name + ".0", keys, f, ts, 3, 1, stats=stats, threads=threads)
    h = ops.relu(h)
    _, logit, _ = _conv_layer(P, name + ".2", keys, h, ts, 3, 1, stats=stats, threads=threads)
    return logit


def synthesis(P, y_keys, y_feats, k, stats=None, threads=1, trace=None):
    """g_s inference path (`model/transforms.py:170-212`).  Returns x_keys, x_feats.
    `trace` (dict) receives per-level logits / masks / keys for parity tests."""
    ts = 8
    keys, f, ts = _conv_layer(P, "g_s.up_1.0", y_keys, y_feats, ts, 5, 1, stats=stats, threads=threads)
    f = _gdn(P, "g_s.up_1.1", f, True)
    keys, f, ts = _conv_layer(P, "g_s.up_1.2", keys, f, ts, 5, 2, transposed=True, stats=stats, threads=threads)
    for lvl, (up, pred) in enumerate((("g_s.up_1", "g_s.predict_1"), ("g_s.up_2", "g_s.predict_2"),
                                      ("g_s.up_3", "g_s.predict_3"))):
        if lvl > 0:
            f = _gdn(P, up + ".0", f, True)
            keys, f, ts = _conv_layer(P, up + ".1", keys, f, ts, 5, 2, transposed=True, stats=stats,
                                      threads=threads)
        logit = _predict(P, pred, keys, f, ts, stats, threads)
        mask = ops.topk_mask(logit[:, 0], k[lvl], _batch_of(keys))
        if trace is not None:
            trace[f"keys_{lvl}"] = keys
            trace[f"feats_{lvl}"] = f
            trace[f"logit_{lvl}"] = logit
            trace[f"mask_{lvl}"] = mask
        keys, f = ops.prune(keys, f, mask)
    _, f, _ = _conv_layer(P, "g_s.color_conv.0", keys, f, ts, 1, 1, stats=stats)
    return keys, f


def hyper_analysis(P, y_keys, y_feats, stats=None, threads=1):
    """h_a (`model/entropy_models.py:177-183`); h_a convs are bias-free (ME default)."""
    pre = "entropy_model.h_a."
    keys, f, ts = _conv_layer(P, pre + "0", y_keys, y_feats, 8, 3, 1, bias=False, stats=stats, threads=threads)
    f = ops.leaky_relu(f)
    keys, f, ts = _conv_layer(P, pre + "2", keys, f, ts, 3, 2, bias=False, stats=stats, threads=threads)
    f = ops.leaky_relu(f)
    keys, f, ts = _conv_layer(P, pre + "4", keys, f, ts, 3, 2, bias=False, stats=stats, threads=threads)
    return keys, f


def hyper_synthesis(P, z_keys, z_hat, stats=None, threads=1):
    """h_s (`model/entropy_models.py:185-191`)."""
    pre = "entropy_model.h_s."
    keys, f, ts = _conv_layer(P, pre + "0", z_keys, z_hat, 32, 2, 2, transposed=True, stats=stats, threads=threads)
    f = ops.leaky_relu(f)
    keys, f, ts = _conv_layer(P, pre + "2", keys, f, ts, 2, 2, transposed=True, stats=stats, threads=threads)
    f = ops.leaky_relu(f)
    keys, f, ts = _conv_layer(P, pre + "4", keys, f, ts, 3, 1, stats=stats, threads=threads)
    return keys, f


def _eb_params(P):
    pre = "entropy_model.entropy_bottleneck."
    return {k[len(pre):]: v for k, v in P.items() if k.startswith(pre)}


def _mlp(P, pre, x, n_layers, final=None):
    """nn.Sequential of Linear/ReLU (+Softplus) (`model/entropy_models.py:193-215`)."""
    h = np.asarray(x, dtype=F32)
    idx = 0
    for li in range(n_layers):
        h = h @ P[f"{pre}.{idx}.weight"].T + P[f"{pre}.{idx}.bias"]
        idx += 2
        if li < n_layers - 1:
            h = np.maximum(h, 0)
    if final == "softplus":
        h = en._softplus(h)
    return h.astype(F32)


def _gain(P, cfg, q, y_keys, n_ch):
    """scale / rescale rows per y element (`model/entropy_models.py:386-393,451-465`)."""
    n = len(y_keys)
    if not cfg.get("adaptive_BN", True):
        one = np.ones((n, n_ch), dtype=F32)
        return one, one
    b = _batch_of(y_keys)
    scale = _mlp(P, "entropy_model.scale_nn", q, 3, "softplus") + F32(1e-4)
    scale_rows = scale[b]
    if cfg.get("inverse_rescaling", False):
        rescale_rows = F32(1.0) / scale_rows
    else:
        rescale_rows = (F32(1.0) / _mlp(P, "entropy_model.rescale_nn", q, 3, "softplus"))[b]
    return scale_rows.astype(F32), rescale_rows.astype(F32)


def _gaussian_params(P, z_keys, z_hat, y_keys, stats, threads):
    g_keys, g = hyper_synthesis(P, z_keys, z_hat, stats, threads)
    g_at_y = ops.features_at(g_keys, g, y_keys)
    c = g_at_y.shape[1] // 2
    return g_at_y[:, :c], g_at_y[:, c:]  # scales_hat, means_hat  (chunk(2, dim=1))


def entropy_compress(P, cfg, y_keys, y_feats, q, stats=None, threads=1):
    """`MeanScaleHyperprior.compress` (`model/entropy_models.py:344-406`) up to the rANS
    boundary: returns integer symbols + everything the coder consumes."""
    z_keys, z = hyper_analysis(P, y_keys, y_feats, stats, threads)
    eb = _eb_params(P)
    z_sym, z_hat = en.eb_quantize(eb, z.T)            # [C,Nz]
    z_lik = en.eb_likelihood(eb, z_hat)
    scales_hat, means_hat = _gaussian_params(P, z_keys, z_hat.T, y_keys, stats, threads)
    scale, _ = _gain(P, cfg, q, y_keys, y_feats.shape[1])
    s = scales_hat * scale
    indexes = en.build_indexes(s)
    y_sym = en.quantize_symbols(y_feats * scale, means_hat * scale)
    y_lik = en.gaussian_likelihood(y_sym.astype(F32), s)   # likelihood of round(y*g - mu*g)
    return dict(y_keys=y_keys, z_keys=z_keys, y_symbols=y_sym, z_symbols=z_sym.T.copy(),
                indexes=indexes, scales_hat=scales_hat, means_hat=means_hat,
                y_likelihood=y_lik, z_likelihood=z_lik.T.copy(), shape=[len(z_keys)])


def entropy_decompress(P, cfg, y_keys, z_keys, y_symbols, z_symbols, q, stats=None, threads=1):
    """`MeanScaleHyperprior.decompress` (`model/entropy_models.py:409-490`) from symbols."""
    eb = _eb_params(P)
    med = en.eb_medians(eb)[:, 0, :]
    z_hat = (np.asarray(z_symbols).T.astype(F32) + med).astype(F32)   # [C,Nz]
    scales_hat, means_hat = _gaussian_params(P, z_keys, z_hat.T, y_keys, stats, threads)
    scale, rescale = _gain(P, cfg, q, y_keys, means_hat.shape[1])
    indexes = en.build_indexes(scales_hat * scale)
    if cfg.get("quantization_offset", False):
        qv = np.asarray(y_symbols).astype(F32)
        q_abs, signs = np.abs(qv), np.sign(qv)
        stdev = en.lower_bound(scales_hat * scale, en.SCALE_BOUND)
        inp = np.stack([scale, stdev], axis=-1)                       # [...,2] = (scale, stddev)
        off = -_mlp(P, "entropy_model.quant_nn", inp, 3)[..., 0]
        off[q_abs < 1e-4] = 0
        y_hat = signs * (q_abs + off) * rescale + means_hat
    else:
        y_hat = en.dequantize(y_symbols, means_hat * scale)           # `entropy_models.py:484`
    return y_hat.astype(F32), indexes


# --------------------------------------------------------------------------
# model level
# --------------------------------------------------------------------------

def partition_blocks(points, block_size):
    """Block partition of `UnifiedModel.compress` (`model/model.py:121-127`)."""
    xyz = np.asarray(points)[:, :3]
    mn = xyz.min(axis=0)
    bi = np.floor((xyz - mn) / block_size).astype(np.int64)
    code = bi[:, 0] * 10 ** 6 + bi[:, 1] * 10 ** 3 + bi[:, 2]
    order = np.argsort(code, kind="stable")
    _, counts = np.unique(code[order], return_counts=True)
    return order, counts


def block_input(x_block):
    """Per-block input tensor (`model/model.py:141-161`): floor coords, dedup keep-first,
    features [1, r, g, b]; returned in canonical key order."""
    xyz = np.floor(np.asarray(x_block)[:, :3]).astype(np.int64)
    C = np.concatenate([np.zeros((len(xyz), 1), dtype=np.int64), xyz], axis=1)
    keys, first = co.canonicalize(C)
    rgb = np.asarray(x_block, dtype=F32)[first, 3:6]
    feats = np.concatenate([np.ones((len(keys), 1), dtype=F32), rgb], axis=1)
    return keys, feats


def compress(P, cfg, pointcloud, q, block_size=1024, stats=None, threads=1):
    """`UnifiedModel.compress(path=None)` (`model/model.py:94-187`) with symbols in place
    of rANS strings.  Returns one dict per block."""
    order, counts = partition_blocks(pointcloud, block_size)
    xs = np.asarray(pointcloud)[order]
    blocks, start = [], 0
    for c in counts.tolist():
        keys, feats = block_input(xs[start:start + c])
        y_keys, y, k = analysis(P, keys, feats, stats, threads)
        rec = entropy_compress(P, cfg["entropy_model"], y_keys, y, q, stats, threads)
        rec["k"] = k
        rec["q"] = q
        rec["n_points"] = len(keys)
        blocks.append(rec)
        start += c
    return blocks


def decompress(P, cfg, blocks, stats=None, threads=1, trace=None):
    """`UnifiedModel.decompress` from components (`model/model.py:191-250`)."""
    outs = []
    for rec in blocks:
        y_keys = rec["y_keys"]
        # z coordinates from two k3 s2 `down_conv`s (`model/model.py:227-229`): coords only
        z_keys = co.stride_keys(co.stride_keys(y_keys, 16), 32)
        y_hat, _ = entropy_decompress(P, cfg["entropy_model"], y_keys, z_keys, rec["y_symbols"],
                                      rec["z_symbols"], rec["q"], stats, threads)
        x_keys, x = synthesis(P, y_keys, y_hat, rec["k"], stats, threads, trace)
        C = co.unpack_keys(x_keys)[:, 1:4].astype(F32)
        col = np.clip(np.rint(x * F32(255)), 0, 255) / F32(255)
        outs.append(np.concatenate([C, col.astype(F32)], axis=1))
    return np.concatenate(outs, axis=0)


def bits(blocks):
    """-sum log2(likelihood) over y and z (`loss.py:77-79`)."""
    t = 0.0
    for r in blocks:
        t += float(-np.log2(r["y_likelihood"].astype(np.float64)).sum())
        t += float(-np.log2(r["z_likelihood"].astype(np.float64)).sum())
    return t


# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------

def conv_shapes(cfg):
    """(state-dict prefix, K, Cin, Cout, has_bias) of every sparse conv of the model."""
    ga, gs, em = cfg["g_a"], cfg["g_s"], cfg["entropy_model"]
    Cb, Ch = em["C_bottleneck"], em["C_hyper_bottleneck"]
    L = [
        ("g_a.down_conv_1.0", 125, ga["C_in"], ga["N1"], True),
        ("g_a.down_conv_2.0", 125, ga["N1"], ga["N2"], True),
        ("g_a.down_conv_3.0", 125, ga["N2"], ga["N3"], True),
        ("g_a.down_conv_3.2", 125, ga["N3"], ga["N4"], True),
        ("g_s.up_1.0", 125, gs["N4"], gs["N3"], True),
        ("g_s.up_1.2", 125, gs["N3"], gs["N2"], True),
        ("g_s.up_2.1", 125, gs["N2"], gs["N1"], True),
        ("g_s.up_3.1", 125, gs["N1"], gs["N1"] // 4, True),
        ("g_s.color_conv.0", 1, gs["N1"] // 4, gs["C_out"], True),
        ("g_s.predict_1.0", 27, gs["N2"], gs["N2"] // 2, True),
        ("g_s.predict_1.2", 27, gs["N2"] // 2, 1, True),
        ("g_s.predict_2.0", 27, gs["N1"], gs["N1"] // 2, True),
        ("g_s.predict_2.2", 27, gs["N1"] // 2, 1, True),
        ("g_s.predict_3.0", 27, gs["N1"] // 4, gs["N4"] // 8, True),
        ("g_s.predict_3.2", 27, gs["N4"] // 8, 1, True),
        ("g_s.down_conv", 27, 1, 1, False),
        ("entropy_model.h_a.0", 27, Cb, Ch, False),
        ("entropy_model.h_a.2", 27, Ch, Ch, False),
        ("entropy_model.h_a.4", 27, Ch, Ch, False),
        ("entropy_model.h_s.0", 8, Ch, Ch, True),
        ("entropy_model.h_s.2", 8, Ch, Cb * 3 // 2, True),
        ("entropy_model.h_s.4", 27, Cb * 3 // 2, Cb * 2, True),
    ]
    return L


def gdn_names(cfg):
    ga, gs = cfg["g_a"], cfg["g_s"]
    return [("g_a.down_conv_1.1", ga["N1"]), ("g_a.down_conv_2.1", ga["N2"]), ("g_a.down_conv_3.1", ga["N3"]),
            ("g_s.up_1.1", gs["N3"]), ("g_s.up_2.0", gs["N2"]), ("g_s.up_3.0", gs["N1"])]


def random_params(cfg, seed=0, gdn_jitter=True, gain=4.0):
    """Seeded random parameter set with the reference's state-dict names and shapes
    (no trained weights ship: `README.md:122`).  Not the product's initialiser; tests copy
    these arrays into the HIP-backed modules so both sides hold identical values."""
    rng = np.random.default_rng(seed)
    P = {}
    for name, K, cin, cout, has_bias in conv_shapes(cfg):
        a = gain / np.sqrt(cin * K)   # gain > 1 keeps activations O(1..10) on sparse supports
        shape = (K, cin, cout) if K > 1 else (cin, cout)
        P[name + ".kernel"] = rng.uniform(-a, a, shape).astype(F32)
        if has_bias:
            P[name + ".bias"] = rng.uniform(-a, a, (1, cout)).astype(F32)
    for name, c in gdn_names(cfg):
        beta = np.sqrt(np.ones(c) + ops.PEDESTAL)
        gamma = np.sqrt(0.1 * np.eye(c) + ops.PEDESTAL)
        if gdn_jitter:  # move off the diagonal init so the [C,C] product is exercised
            beta = beta + rng.uniform(0, 0.2, c)
            gamma = gamma + rng.uniform(0, 0.05, (c, c))
        P[name + ".beta"] = beta.astype(F32)
        P[name + ".gamma"] = gamma.astype(F32)
    em = cfg["entropy_model"]
    Cb, Ch = em["C_bottleneck"], em["C_hyper_bottleneck"]
    eb = en.eb_init(Ch, seed=seed + 1)
    for k, v in eb.items():
        P["entropy_model.entropy_bottleneck." + k] = v
    def lin(pre, dims):
        idx = 0
        for i in range(len(dims) - 1):
            a = 1.0 / np.sqrt(dims[i])
            P[f"{pre}.{idx}.weight"] = rng.uniform(-a, a, (dims[i + 1], dims[i])).astype(F32)
            P[f"{pre}.{idx}.bias"] = rng.uniform(-a, a, (dims[i + 1],)).astype(F32)
            idx += 2
    lin("entropy_model.scale_nn", [2, 8, Cb // 4, Cb])
    lin("entropy_model.rescale_nn", [2, 8, Cb // 4, Cb])
    lin("entropy_model.quant_nn", [2, 10, 10, 1])
    return P


R2_CONFIG = {
    # `configs/CVPR_inverse_scaling_fixed_R2.yaml:6-26`
    "entropy_model": dict(C_bottleneck=128, C_hyper_bottleneck=192, quantization_mode="ste",
                          inverse_rescaling=False, quantization_offset=False,
                          entropy_bottleneck_vbr=False, adaptive_BN=False),
    "g_a": dict(C_in=4, N1=128, N2=128, N3=128, N4=128),
    "g_s": dict(C_out=3, N1=128, N2=128, N3=128, N4=128),
}


def small_config(n=16, cb=64, ch=32, adaptive=False, offsets=False, inverse=False):
    """Narrow variant of the same architecture for fast tests (channel counts the MFMA tiling takes:
    4/8/16 or multiples of 32, including C_bottleneck*3//2)."""
    return {
        "entropy_model": dict(C_bottleneck=cb, C_hyper_bottleneck=ch, quantization_mode="ste",
                              inverse_rescaling=inverse, quantization_offset=offsets,
                              entropy_bottleneck_vbr=False, adaptive_BN=adaptive),
        "g_a": dict(C_in=4, N1=n, N2=n, N3=n, N4=cb),
        "g_s": dict(C_out=3, N1=n, N2=n, N3=n, N4=cb),
    }
