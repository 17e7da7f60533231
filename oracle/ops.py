"""Feature-side operators of the oracle (numpy fp32).  Test infrastructure.

Restates the MinkowskiEngine / CompressAI operator semantics the reference
calls; every function cites the reference
call site it stands in for.
"""
import numpy as np

F32 = np.float32


def conv(feats, W, bias, nbr, chunk=1 << 20):
    """Sparse convolution given a neighbour table.

    out[o] = bias + sum_k feats[nbr[k,o]] @ W[k]   (cross-correlation, no flip)
    Stands in for `ME.MinkowskiConvolution` / `MinkowskiGenerativeConvolutionTranspose`
    forward (`model/transforms.py:33-43,127-166`, `model/entropy_models.py:178-190`).
    W: [K,Cin,Cout] (or [Cin,Cout] for K=1), bias: [1,Cout] / [Cout] / None.
    Accumulation order: k ascending (float32 adds)."""
    feats = np.ascontiguousarray(feats, dtype=F32)
    W = np.asarray(W, dtype=F32)
    if W.ndim == 2:
        W = W[None]
    K, Cin, Cout = W.shape
    assert nbr.shape[0] == K and feats.shape[1] == Cin
    n_out = nbr.shape[1]
    out = np.zeros((n_out, Cout), dtype=F32)
    if bias is not None:
        out += np.asarray(bias, dtype=F32).reshape(1, Cout)
    for k in range(K):
        o = np.nonzero(nbr[k] >= 0)[0]
        for s in range(0, o.size, chunk):
            oo = o[s:s + chunk]
            out[oo] += feats[nbr[k, oo]] @ W[k]
    return out


def pair_count(nbr):
    """P = number of (in,out) pairs of a kernel map (FLOP = 2*P*Cin*Cout)."""
    return int((nbr >= 0).sum())


def relu(x):
    """`ME.MinkowskiReLU` (`model/transforms.py:148,153,158`)."""
    return np.maximum(x, F32(0))


def leaky_relu(x, slope=0.01):
    """`ME.MinkowskiLeakyReLU`, negative_slope 0.01 (`model/entropy_models.py:179,181,187,189`)."""
    x = np.asarray(x, dtype=F32)
    return np.where(x >= 0, x, x * F32(slope)).astype(F32)


PEDESTAL = 2.0 ** -36


def nonneg_reparam(x, minimum=0.0):
    """CompressAI `NonNegativeParametrizer.forward`:
    max(x, sqrt(minimum + pedestal))^2 - pedestal, in float32."""
    bound = F32((minimum + PEDESTAL) ** 0.5)
    x = np.maximum(np.asarray(x, dtype=F32), bound)
    return (x * x - F32(PEDESTAL)).astype(F32)


def gdn(feats, beta_raw, gamma_raw, inverse=False, beta_min=1e-6):
    """`MinkowskiGDN.forward` (`model/blocks.py:38-57`): GDN1 form,
    norm = beta + |x| @ gamma^T ; y = x / norm (GDN) or x * norm (IGDN)."""
    beta = nonneg_reparam(beta_raw, beta_min)
    gamma = nonneg_reparam(gamma_raw, 0.0)
    x = np.asarray(feats, dtype=F32)
    norm = np.abs(x) @ gamma.T + beta[None, :]
    if not inverse:
        norm = F32(1.0) / norm
    return (x * norm).astype(F32)


def topk_mask(logits, k, batch=None):
    """`SparseSynthesisTransform._topk_prediction` (`model/transforms.py:228-254`).

    Per batch index b the k[b] largest logits are kept.  torch.topk leaves ties
    unspecified; the build fixes the total order (logit descending, canonical
    row ascending) -- SURVEY A.7.  Rows are assumed to be in canonical order."""
    logits = np.asarray(logits, dtype=F32).reshape(-1)
    mask = np.zeros(logits.shape[0], dtype=bool)
    if batch is None:
        batch = np.zeros(logits.shape[0], dtype=np.int64)
    ks = list(np.atleast_1d(k))
    for b in np.unique(batch):
        rows = np.nonzero(batch == b)[0]
        kb = int(ks[int(b)])
        kb = min(kb, rows.size)
        order = np.argsort(-logits[rows], kind="stable")
        mask[rows[order[:kb]]] = True
    return mask


def prune(keys, feats, mask):
    """`ME.MinkowskiPruning` (`model/transforms.py:163,257-282`)."""
    return keys[mask], feats[mask]


def features_at(keys, feats, query_keys):
    """`SparseTensor.features_at_coordinates` for on-grid queries (SURVEY A.8;
    `model/entropy_models.py:294,381,446`): the row at the query or zeros."""
    from .coords import lookup
    idx = lookup(keys, query_keys)
    out = np.zeros((len(query_keys), feats.shape[1]), dtype=F32)
    hit = idx >= 0
    out[hit] = feats[idx[hit]]
    return out
