"""Pins the oracle's sparse operators against PyTorch dense ops:
sparse conv on a voxel set == dense conv3d on the zero-filled grid sampled at the output
set; generative transpose == conv_transpose3d support set.  The reference holds no
fixtures for this path ("parity unpinned"), so this independent equivalence is the pin."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import coords as co
from oracle import ops
from oracle import codec


def _cloud(seed, size, p, cin, margin=0):
    rng = np.random.default_rng(seed)
    occ = rng.random((size, size, size)) < p
    xyz = np.argwhere(occ) + margin           # columns (x,y,z)
    C = np.concatenate([np.zeros((len(xyz), 1), dtype=np.int64), xyz], axis=1)
    keys, first = co.canonicalize(C)
    feats = rng.standard_normal((len(keys), cin)).astype(np.float32)
    return keys, feats


def _dense(keys, feats, grid, ts=1):
    C = co.unpack_keys(keys)
    d = torch.zeros(1, feats.shape[1], grid, grid, grid, dtype=torch.float64)
    g = C[:, 1:] // ts
    d[0, :, g[:, 2], g[:, 1], g[:, 0]] = torch.from_numpy(feats.T.astype(np.float64))  # dims (z,y,x)
    return d


def _dense_weight(W, k, transposed=False):
    K, cin, cout = W.shape
    w = torch.from_numpy(W.astype(np.float64)).view(k, k, k, cin, cout)  # (kz,ky,kx,ci,co): x fastest
    return w.permute(3, 4, 0, 1, 2).contiguous() if transposed else w.permute(4, 3, 0, 1, 2).contiguous()


def _sample(dense, keys, ts=1):
    C = co.unpack_keys(keys)
    g = C[:, 1:] // ts
    return dense[0][:, g[:, 2], g[:, 1], g[:, 0]].T.numpy()


@pytest.mark.parametrize("k,stride,ts", [(3, 1, 1), (5, 1, 1), (5, 2, 1), (3, 2, 2), (5, 1, 4), (1, 1, 1)])
def test_conv_matches_dense(k, stride, ts):
    size, cin, cout = 12, 3, 5
    keys, feats = _cloud(1, size, 0.2, cin)
    # place the cloud on a stride-ts lattice
    C = co.unpack_keys(keys).astype(np.int64)
    C[:, 1:] *= ts
    keys = co.pack_keys(C)
    rng = np.random.default_rng(2)
    K = k ** 3
    W = rng.standard_normal((K, cin, cout)).astype(np.float32)
    b = rng.standard_normal((1, cout)).astype(np.float32)
    out_keys = keys if stride == 1 else co.stride_keys(keys, ts * stride)
    nbr = co.kernel_map(keys, out_keys, k, ts)
    got = ops.conv(feats, W, b, nbr)
    # pair form agrees with table form
    pairs = codec.kernel_map_pairs(keys, out_keys, k, ts)
    assert np.array_equal(codec.pairs_to_nbr(pairs, len(out_keys)), nbr)
    dense = F.conv3d(_dense(keys, feats, size, ts), _dense_weight(W, k), bias=torch.from_numpy(b[0].astype(np.float64)),
                     padding=(k - 1) // 2)
    want = _sample(dense, out_keys, ts)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
    if stride > 1:  # output set == occupied coarse cells
        Cg = co.unpack_keys(keys)[:, 1:] // (ts * stride)
        assert len(out_keys) == len(np.unique(Cg, axis=0))


@pytest.mark.parametrize("k,pad", [(5, 2), (2, 0)])
def test_generative_transpose_matches_dense(k, pad):
    size, cin, cout, ts_in = 7, 4, 3, 2
    margin = 2
    keys, feats = _cloud(3, size, 0.25, cin, margin=margin)
    C = co.unpack_keys(keys).astype(np.int64)
    C[:, 1:] *= ts_in
    keys = co.pack_keys(C)
    rng = np.random.default_rng(4)
    W = rng.standard_normal((k ** 3, cin, cout)).astype(np.float32)
    b = rng.standard_normal((1, cout)).astype(np.float32)
    ts_out = ts_in // 2
    out_keys = co.expand_keys(keys, k, ts_out)
    pairs = codec.kernel_map_pairs(keys, out_keys, k, ts_out, transposed=True)
    assert sum(len(i) for i, _ in pairs) == len(keys) * k ** 3      # every (in,k) is one pair
    nbr = co.kernel_map(keys, out_keys, k, ts_out, transposed=True)
    assert np.array_equal(codec.pairs_to_nbr(pairs, len(out_keys)), nbr)
    got = ops.conv(feats, W, b, nbr)
    grid = size + 2 * margin
    dense = F.conv_transpose3d(_dense(keys, feats, grid, ts_in), _dense_weight(W, k, True), stride=2, padding=pad)
    support = F.conv_transpose3d((_dense(keys, np.ones((len(keys), 1), np.float32), grid, ts_in)),
                                 torch.ones(1, 1, k, k, k, dtype=torch.float64), stride=2, padding=pad)
    want = _sample(dense, out_keys, ts_out) + b
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
    assert int((support > 0).sum()) == len(out_keys)               # same support set


def test_negative_coordinates_and_order():
    C = np.array([[0, 0, 0, 0], [0, -2, 5, 1], [1, 0, 0, 0], [0, 0, 0, -1], [0, 0, 0, 0]])
    keys, first = co.canonicalize(C)
    U = co.unpack_keys(keys)
    assert U.tolist() == [[0, -2, 5, 1], [0, 0, 0, -1], [0, 0, 0, 0], [1, 0, 0, 0]]
    assert first.tolist() == [1, 3, 0, 2]                            # first occurrence wins
    # floor semantics for negatives in stride
    assert co.unpack_keys(co.stride_keys(keys, 2)).tolist() == [[0, -2, 4, 0], [0, 0, 0, -2], [0, 0, 0, 0], [1, 0, 0, 0]]


def test_kernel_offset_enumeration():
    o = co.kernel_offsets(3)
    assert o[0].tolist() == [-1, -1, -1] and o[1].tolist() == [0, -1, -1] and o[3].tolist() == [-1, 0, -1]
    assert o[13].tolist() == [0, 0, 0] and co.kernel_offsets(2)[1].tolist() == [1, 0, 0]


def test_gdn_matches_conv1d():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((50, 6)).astype(np.float32)
    beta = (1 + rng.random(6)).astype(np.float32)
    gamma = (0.3 * rng.random((6, 6)) + 0.3 * np.eye(6)).astype(np.float32)
    for inverse in (False, True):
        got = ops.gdn(x, beta, gamma, inverse)
        b = torch.from_numpy(ops.nonneg_reparam(beta, 1e-6))
        g = torch.from_numpy(ops.nonneg_reparam(gamma)).reshape(6, 6, 1)
        xt = torch.from_numpy(x).T.unsqueeze(0)
        norm = F.conv1d(xt.abs(), g, b)[0].T
        want = (torch.from_numpy(x) * (norm if inverse else 1.0 / norm)).numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def test_topk_tiebreak_and_prune():
    logits = np.array([0.5, 1.0, 0.5, 0.5, 2.0], dtype=np.float32)
    m = ops.topk_mask(logits, [3])
    assert m.tolist() == [True, True, False, False, True]           # tie -> lowest canonical row
    keys = np.arange(5, dtype=np.int64)
    k2, f2 = ops.prune(keys, logits[:, None], m)
    assert k2.tolist() == [0, 1, 4]
