"""Coordinate-set arithmetic of the oracle (numpy, int64).  Test infrastructure.

Restates what the reference obtains from MinkowskiEngine's coordinate manager
(reference call sites `model/model.py:142-161,227-229`,
`model/transforms.py:32-44,126-166`, `model/entropy_models.py:177-191`).

Canonical form: a coordinate set is a strictly ascending int64 array of packed
keys  key = b<<48 | (x+2^15)<<32 | (y+2^15)<<16 | (z+2^15).  Ascending key order
equals lexicographic (b,x,y,z) order, i.e. the order `utils.sort_tensor`
(`utils.py:142-165`) produces.
"""
import numpy as np

BIAS = 1 << 15
FIELD = 0xFFFF


def pack_keys(C):
    """[N,4] integer (b,x,y,z) -> int64 keys."""
    C = np.asarray(C).astype(np.int64)
    assert C.ndim == 2 and C.shape[1] == 4
    if C.size:
        assert C[:, 0].min() >= 0 and C[:, 0].max() < (1 << 15), "batch index out of range"
        assert C[:, 1:].min() >= -BIAS and C[:, 1:].max() < BIAS, "coordinate out of 16-bit range"
    return (C[:, 0] << 48) | ((C[:, 1] + BIAS) << 32) | ((C[:, 2] + BIAS) << 16) | (C[:, 3] + BIAS)


def unpack_keys(keys):
    """int64 keys -> [N,4] int32 (b,x,y,z)."""
    keys = np.asarray(keys, dtype=np.int64)
    out = np.empty((keys.shape[0], 4), dtype=np.int32)
    out[:, 0] = keys >> 48
    out[:, 1] = ((keys >> 32) & FIELD) - BIAS
    out[:, 2] = ((keys >> 16) & FIELD) - BIAS
    out[:, 3] = (keys & FIELD) - BIAS
    return out


def canonicalize(C):
    """Sorted unique keys + `first` (canonical position -> first user row holding it).

    ME.SparseTensor keeps the first occurrence of a duplicated coordinate
    (`model/model.py:152-156`)."""
    keys = pack_keys(C)
    uniq, first = np.unique(keys, return_index=True)
    return uniq, first.astype(np.int64)


def sparse_quantize(C, F=None, quantization_size=1.0):
    """`ME.utils.sparse_quantize` (`model/model.py:152-156`):
    floor(c/qs), keep first occurrence, ORIGINAL relative order."""
    q = np.floor(np.asarray(C, dtype=np.float64) / quantization_size).astype(np.int32)
    keys = pack_keys(q)
    _, first = np.unique(keys, return_index=True)
    first = np.sort(first)
    if F is None:
        return q[first]
    return q[first], np.asarray(F)[first]


def stride_keys(keys, new_stride):
    """Output set of a strided conv: unique(floor(c/m)*m), m=new tensor stride."""
    C = unpack_keys(keys).astype(np.int64)
    C[:, 1:] = np.floor_divide(C[:, 1:], new_stride) * new_stride
    return np.unique(pack_keys(C))


def kernel_offsets(kernel_size):
    """Kernel region: odd k -> -(k-1)/2..(k-1)/2, even k -> 0..k-1;
    offset index kidx = ix + k*iy + k*k*iz (x fastest).  Returns [K,3] int64 (dx,dy,dz)."""
    k = int(kernel_size)
    ax = np.arange(k) - (k - 1) // 2 if k % 2 == 1 else np.arange(k)
    dz, dy, dx = np.meshgrid(ax, ax, ax, indexing="ij")
    return np.stack([dx.ravel(), dy.ravel(), dz.ravel()], axis=1).astype(np.int64)


def offset_deltas(offsets, step):
    """Packed-key delta of each (dx,dy,dz)*step.  Adding it to a key moves the
    coordinate as long as every biased field stays inside [0, 2^16)."""
    o = np.asarray(offsets, dtype=np.int64) * int(step)
    return (o[:, 0] << 32) + (o[:, 1] << 16) + o[:, 2]


def expand_keys(keys, kernel_size, out_stride):
    """Output set of a generative transposed conv:
    unique{ c + off_k*ts_out }."""
    d = offset_deltas(kernel_offsets(kernel_size), out_stride)
    C = unpack_keys(keys).astype(np.int64)
    o = kernel_offsets(kernel_size) * int(out_stride)
    lo = C[:, 1:].min(axis=0) + o.min(axis=0) if len(C) else 0
    hi = C[:, 1:].max(axis=0) + o.max(axis=0) if len(C) else 0
    assert np.all(lo >= -BIAS) and np.all(hi < BIAS)
    cand = (np.asarray(keys, dtype=np.int64)[:, None] + d[None, :]).ravel()
    return np.unique(cand)


def lookup(sorted_keys, queries):
    """Row index of each query key in a canonical set, -1 when absent."""
    sorted_keys = np.asarray(sorted_keys, dtype=np.int64)
    queries = np.asarray(queries, dtype=np.int64)
    if sorted_keys.size == 0:
        return np.full(queries.shape, -1, dtype=np.int32)
    pos = np.searchsorted(sorted_keys, queries)
    pos_c = np.minimum(pos, sorted_keys.size - 1)
    hit = sorted_keys[pos_c] == queries
    return np.where(hit, pos_c, -1).astype(np.int32)


def kernel_map(in_keys, out_keys, kernel_size, step, transposed=False):
    """Neighbour table nbr[K, N_out].

    conv:       nbr[k,o] = i  with C_in[i]  = C_out[o] + off_k*step   (step = ts_in)
    transposed: nbr[k,o] = i  with C_out[o] = C_in[i]  + off_k*step   (step = ts_out)
    -1 where no such input row exists."""
    d = offset_deltas(kernel_offsets(kernel_size), step)
    if transposed:
        d = -d
    out_keys = np.asarray(out_keys, dtype=np.int64)
    nbr = np.empty((d.shape[0], out_keys.shape[0]), dtype=np.int32)
    for k in range(d.shape[0]):
        nbr[k] = lookup(in_keys, out_keys + d[k])
    return nbr
