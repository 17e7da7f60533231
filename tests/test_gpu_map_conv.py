"""a2 / a3 / a5: kernel maps (bit exact) and convolution / GDN features (1e-4) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import coords as co, ops, codec
from tests.util import dev, t, n, cloud_keys, assert_close

pytestmark = pytest.mark.gpu


def _cs(keys, ts, batch=1):
    from unified_point_cloud_compression_amd import sparse as S
    C = co.unpack_keys(keys)
    return S.CoordSet(t(keys), len(keys), ts, S.Bounds(batch - 1, C[:, 1:].min(0), C[:, 1:].max(0)))


@pytest.fixture(params=[True, False], ids=["grid", "bsearch"])
def lookup_mode(request):
    """Both neighbour-lookup structures (bitmap+rank grid index / binary search) must give identical maps."""
    from unified_point_cloud_compression_amd import sparse as S
    old = S.USE_GRID
    S.USE_GRID = request.param
    yield request.param
    S.USE_GRID = old


@pytest.mark.parametrize("ks,stride,ts", [(3, 1, 1), (5, 1, 8), (5, 2, 1), (3, 2, 8), (5, 2, 4)])
def test_conv_map_bit_exact(ks, stride, ts, lookup_mode):
    keys = cloud_keys(ks + stride, 28, 0.12, ts, batch=2)
    C = co.unpack_keys(keys)
    C[:, 1:] -= 2 * ts                                   # negative coordinates too
    keys = np.unique(co.pack_keys(C))
    cs = _cs(keys, ts, 2)
    out = cs if stride == 1 else cs.stride(ts * stride)
    m = cs.kernel_map(out, ks)
    out_keys = keys if stride == 1 else co.stride_keys(keys, ts * stride)
    want = co.kernel_map(keys, out_keys, ks, ts)
    assert np.array_equal(n(m.dense()), want)
    assert cs.kernel_map(out, ks) is m                    # cached per (in set, out set, kernel)
    mz = cs.kernel_map(out, ks, morton=True)              # Z-curve visiting order: same map, positions permuted
    assert np.array_equal(n(mz.dense()), want)
    rows = n(mz.rows)[:out.n]
    assert np.array_equal(np.sort(rows), np.arange(out.n))
    C = co.unpack_keys(out_keys)[rows].astype(np.int64)
    code = C[:, 0] << 48
    sh = int(np.log2(ts))
    for bit in range(16):
        for ax, pos in ((3, 0), (2, 1), (1, 2)):                     # z lowest, then y, then x
            cell = (C[:, ax] + (1 << 15)) >> sh
            code |= ((cell >> bit) & 1) << (3 * bit + pos)
    assert np.all(np.diff(code) > 0)                      # positions ascend along the Z-curve (batch major)


@pytest.mark.parametrize("ks,ts_in", [(5, 2), (2, 2), (2, 32), (5, 8)])
def test_transposed_map_bit_exact(ks, ts_in, lookup_mode):
    keys = cloud_keys(3, 18, 0.1, ts_in, batch=2)
    cs = _cs(keys, ts_in, 2)
    ts_out = ts_in // 2
    out = cs.expand(ks, ts_out)
    m = cs.kernel_map(out, ks, transposed=True, up_stride=2)
    out_keys = co.expand_keys(keys, ks, ts_out)
    want = co.kernel_map(keys, out_keys, ks, ts_out, transposed=True)
    dense = n(m.dense())
    assert np.array_equal(dense, want)
    assert (dense >= 0).sum() == len(keys) * ks ** 3      # every (in, k) is exactly one pair
    rows = n(m.rows)[:out.n]
    assert np.array_equal(np.sort(rows), np.arange(out.n))  # positions are a permutation of the output rows


SHAPES = [(4, 128), (128, 128), (128, 64), (64, 1), (128, 32), (32, 16), (16, 1), (32, 3), (128, 192), (192, 256),
          (1, 1), (8, 8), (16, 32), (4, 2)]


@pytest.mark.parametrize("cin,cout", SHAPES)
def test_conv_features_k3(cin, cout):
    from unified_point_cloud_compression_amd import sparse as S, lib as L
    keys = cloud_keys(cin * 7 + cout, 20, 0.15, 1)
    cs = _cs(keys, 1)
    rng = np.random.default_rng(cin + cout)
    f = rng.standard_normal((len(keys), cin)).astype(np.float32)
    W = (rng.standard_normal((27, cin, cout)) / np.sqrt(cin * 8)).astype(np.float32)
    b = rng.standard_normal((1, cout)).astype(np.float32)
    m = cs.kernel_map(cs, 3, morton=(cin + cout) % 2 == 1)     # half of the shapes run on a Z-curve ordered map
    pk = S.PackedConv().get(torch.nn.Parameter(t(W)))
    for act, fn in ((L.ACT_NONE, lambda v: v), (L.ACT_RELU, ops.relu), (L.ACT_LEAKY, ops.leaky_relu)):
        got = S.conv_forward(t(f), pk, t(b), 27, cin, cout, m, len(keys), act)
        want = fn(ops.conv(f, W, b, co.kernel_map(keys, keys, 3, 1)))
        assert_close(n(got), want, what=f"conv {cin}->{cout} act={act}")


@pytest.mark.parametrize("cin,cout,ks,stride", [(4, 128, 5, 2), (128, 128, 5, 2), (128, 128, 5, 1), (192, 192, 3, 2)])
def test_conv_strided_k5(cin, cout, ks, stride):
    from unified_point_cloud_compression_amd import sparse as S
    keys = cloud_keys(11, 26, 0.1, 1, batch=2)
    cs = _cs(keys, 1, 2)
    out = cs if stride == 1 else cs.stride(stride)
    rng = np.random.default_rng(5)
    f = rng.standard_normal((len(keys), cin)).astype(np.float32)
    K = ks ** 3
    W = (rng.standard_normal((K, cin, cout)) / np.sqrt(cin * 10)).astype(np.float32)
    m = cs.kernel_map(out, ks)
    got = S.conv_forward(t(f), S.PackedConv().get(torch.nn.Parameter(t(W))), None, K, cin, cout, m, out.n)
    out_keys = keys if stride == 1 else co.stride_keys(keys, stride)
    want = ops.conv(f, W, None, co.kernel_map(keys, out_keys, ks, 1))
    assert_close(n(got), want, what="strided conv")


@pytest.mark.parametrize("ks,stride,cout,act", [(5, 2, 128, "none"), (5, 1, 128, "relu"), (3, 2, 256, "leaky")])
def test_four_channel_input_layer_flattened_form(ks, stride, cout, act):
    """`k_conv_in4_bf` (the codec's first layer: 4 input channels, the (offset, channel) pairs flattened into one 32-wide-chunk
    reduction, six bf16 terms) against the oracle and against the offset-by-offset kernel it replaces on large inputs; two
    batches, a ragged last tile, K * 4 not a multiple of 32 (5^3 -> 500, 3^3 -> 108), bias and fused activations."""
    from unified_point_cloud_compression_amd import sparse as S, lib as L
    keys = cloud_keys(17 + ks, 30, 0.12, 1, batch=2)
    cs = _cs(keys, 1, 2)
    out = cs if stride == 1 else cs.stride(stride)
    rng = np.random.default_rng(ks)
    f = np.concatenate([np.ones((len(keys), 1)), rng.random((len(keys), 3)) * 255], axis=1).astype(np.float32)   # [1, r, g, b]
    K = ks ** 3
    W = (rng.standard_normal((K, 4, cout)) / 40).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    m = cs.kernel_map(out, ks)
    code = {"none": L.ACT_NONE, "relu": L.ACT_RELU, "leaky": L.ACT_LEAKY}[act]
    fn = {"none": lambda v: v, "relu": lambda v: np.maximum(v, 0), "leaky": lambda v: np.where(v > 0, v, 0.01 * v)}[act]
    packed = S.PackedConv().get(torch.nn.Parameter(t(W)))
    ft, bt = t(f), t(b)
    try:
        L.call("pcc_set_in4_min_rows", 0)
        got = S.conv_forward(ft, packed, bt, K, 4, cout, m, out.n, act=code)
        L.call("pcc_set_in4_min_rows", -1)
        old = S.conv_forward(ft, packed, bt, K, 4, cout, m, out.n, act=code)
    finally:
        L.call("pcc_set_in4_min_rows", 65536)
    out_keys = keys if stride == 1 else co.stride_keys(keys, stride)
    want = fn(ops.conv(f, W, b, co.kernel_map(keys, out_keys, ks, 1)))
    scale = float(np.abs(want).max())
    assert_close(n(got) / scale, want / scale, what="flattened 4-channel convolution vs oracle")
    assert_close(n(got) / scale, n(old) / scale, atol=2e-6, rtol=0, what="flattened form vs the offset-by-offset kernel")


@pytest.mark.parametrize("cin,cout,stride,bias,act", [(128, 128, 2, True, 1), (64, 64, 1, False, 0), (192, 256, 1, True, 2),
                                                       (128, 32, 2, False, 0)])
def test_conv_pair_list_form(cin, cout, stride, bias, act):
    """5x5x5 convolution on a sparse set through the pair-list form (compacted per-offset pair tiles + ordered reduce)
    and through the output-stationary kernel: both against the oracle, and the plan's bookkeeping against the map."""
    from unified_point_cloud_compression_amd import sparse as S
    keys = cloud_keys(21, 30, 0.03, 1, batch=2)
    cs = _cs(keys, 1, 2)
    out = cs if stride == 1 else cs.stride(stride)
    rng = np.random.default_rng(15)
    f = rng.standard_normal((len(keys), cin)).astype(np.float32)
    W = (rng.standard_normal((125, cin, cout)) / np.sqrt(cin * 10)).astype(np.float32)
    b = rng.standard_normal((1, cout)).astype(np.float32) if bias else None
    m = cs.kernel_map(out, 5)
    plan = m.pair_plan()
    assert plan is not None, "map expected to be sparse enough for the pair form"
    pos, pair_in, tile_k, info, padded = plan
    dense = n(m.dense())
    padded_i, tiles, pairs = (int(v) for v in n(info))
    assert pairs == int((dense >= 0).sum()) and padded_i == padded == tiles * 128
    pos_h, pin_h, tk_h = n(pos).reshape(125, out.n), n(pair_in), n(tile_k)
    assert np.array_equal(pos_h >= 0, dense >= 0)
    assert np.array_equal(pin_h[pos_h[dense >= 0]], dense[dense >= 0])          # pair row -> input row
    assert (pin_h >= 0).sum() == pairs                                            # everything else is padding
    kk = np.repeat(np.arange(125), out.n).reshape(125, out.n)
    assert np.array_equal(tk_h[pos_h[dense >= 0] // 128], kk[dense >= 0])        # one offset per 128-pair tile
    pk = S.PackedConv().get(torch.nn.Parameter(t(W)))
    bt = t(b) if bias else None
    got_pairs = S.conv_forward(t(f), pk, bt, 125, cin, cout, m, out.n, act=act, slope=0.2)
    old = S.PAIR_MIN_K
    S.PAIR_MIN_K = 1 << 30
    try:
        got_os = S.conv_forward(t(f), pk, bt, 125, cin, cout, m, out.n, act=act, slope=0.2)
    finally:
        S.PAIR_MIN_K = old
    out_keys = keys if stride == 1 else co.stride_keys(keys, stride)
    want = ops.conv(f, W, b, co.kernel_map(keys, out_keys, 5, 1))
    want = {0: lambda v: v, 1: ops.relu, 2: lambda v: ops.leaky_relu(v, 0.2)}[act](want)
    assert_close(n(got_pairs), want, what="pair-list conv")
    assert_close(n(got_os), want, what="output-stationary conv")


@pytest.mark.parametrize("cin,cout,ks", [(128, 128, 5), (128, 32, 5), (192, 192, 2), (16, 4, 5), (8, 16, 2)])
def test_generative_transpose_features(cin, cout, ks):
    from unified_point_cloud_compression_amd import sparse as S
    keys = cloud_keys(2, 14, 0.1, 2)
    cs = _cs(keys, 2)
    out = cs.expand(ks, 1)
    rng = np.random.default_rng(6)
    f = rng.standard_normal((len(keys), cin)).astype(np.float32)
    K = ks ** 3
    W = (rng.standard_normal((K, cin, cout)) / np.sqrt(cin * 4)).astype(np.float32)
    b = rng.standard_normal((1, cout)).astype(np.float32)
    m = cs.kernel_map(out, ks, transposed=True, up_stride=2)
    got = S.convt_forward(t(f), S.PackedConv(True).get(torch.nn.Parameter(t(W))), t(b), K, cin, cout, m, out.n)
    # the output-stationary kernel on the same transposed map must agree as well
    got2 = S.conv_forward(t(f), S.PackedConv().get(torch.nn.Parameter(t(W))), t(b), K, cin, cout, m, out.n)
    out_keys = co.expand_keys(keys, ks, 1)
    pairs = codec.kernel_map_pairs(keys, out_keys, ks, 1, transposed=True)
    want = codec.conv_pairs(f, W, b, pairs, len(out_keys))
    assert_close(n(got), want, what="generative transpose (input stationary)")
    csr = cs.csr_map(ks, 1)
    assert csr is not None
    got3 = S.convt_forward_csr(t(f), S.PackedConv(True).get(torch.nn.Parameter(t(W))), t(b), K, cin, cout, csr, out.n)
    assert_close(n(got3), want, what="generative transpose (CSR pair lists)")
    assert_close(n(got2), want, what="generative transpose (output stationary)")


@pytest.mark.parametrize("ks", [2, 3, 5])
def test_generative_transpose_stride1_csr(ks):
    """`ME.MinkowskiGenerativeConvolutionTranspose(stride=1)`: input pitch == output pitch, so EVERY kernel offset is a
    compatible source of an output row (ks per axis, not ceil(ks / 2) as when up-sampling by 2).  The CSR pair lists built
    through the grid index must hold all of them (round-2 advisor finding: the count pass probed (ks+1)/2 cells per axis)."""
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    from unified_point_cloud_compression_amd import sparse as S
    keys = cloud_keys(11 + ks, 12, 0.15, 2, batch=2)
    cs = _cs(keys, 2, 2)
    out = cs.expand(ks, 2)
    out_keys = co.expand_keys(keys, ks, 2)
    assert np.array_equal(n(out.keys[:out.n]), out_keys)
    first, pair_ids = cs.csr_map(ks, 2)
    first, pair_ids = n(first), n(pair_ids)
    K = ks ** 3
    assert first[out.n] == len(keys) * K                      # every (input row, offset) is exactly one pair
    want = co.kernel_map(keys, out_keys, ks, 2, transposed=True)          # [K, n_out] input row or -1
    for o in (0, 1, out.n // 2, out.n - 1):
        ids = pair_ids[first[o]:first[o + 1]]
        assert np.array_equal(np.sort(ids), np.sort(np.array([want[k, o] * K + k for k in range(K) if want[k, o] >= 0])))
        assert np.all(np.diff(ids // K) >= 0)                  # ascending input row
    cin, cout = 32, 32
    rng = np.random.default_rng(4)
    f = rng.standard_normal((len(keys), cin)).astype(np.float32)
    layer = ME.MinkowskiGenerativeConvolutionTranspose(cin, cout, kernel_size=ks, stride=1, bias=True, dimension=3).to(dev())
    x = ME.SparseTensor(coordinates=t(co.unpack_keys(keys)), features=t(f), tensor_stride=2)
    with torch.no_grad():
        y = layer(x)
    W, b = n(layer.kernel).reshape(K, cin, cout), n(layer.bias)
    pairs = codec.kernel_map_pairs(keys, out_keys, ks, 2, transposed=True)
    assert_close(n(y.F), codec.conv_pairs(f, W, b, pairs, len(out_keys)), what="stride-1 generative transpose")


def test_conv_1x1_and_row_tails():
    """K=1 needs no map; row counts that are not multiples of the 128-row tile."""
    from unified_point_cloud_compression_amd import sparse as S
    rng = np.random.default_rng(9)
    for rows in (1, 127, 129, 1000):
        for cin, cout in ((32, 3), (128, 128), (16, 16)):
            f = rng.standard_normal((rows, cin)).astype(np.float32)
            W = rng.standard_normal((1, cin, cout)).astype(np.float32)
            b = rng.standard_normal((1, cout)).astype(np.float32)
            got = S.conv_forward(t(f), S.PackedConv().get(torch.nn.Parameter(t(W[0]))), t(b), 1, cin, cout, None, rows)
            assert_close(n(got), f @ W[0] + b, what=f"1x1 {rows}x{cin}->{cout}")


@pytest.mark.parametrize("c", [128, 32, 16, 64])
@pytest.mark.parametrize("inverse", [False, True])
def test_gdn(c, inverse):
    from unified_point_cloud_compression_amd.model.blocks import MinkowskiGDN
    rng = np.random.default_rng(c)
    x = rng.standard_normal((777, c)).astype(np.float32) * 3
    g = MinkowskiGDN(c, inverse=inverse).to(dev()).eval()
    beta = (np.sqrt(1 + ops.PEDESTAL) + rng.uniform(0, 0.3, c)).astype(np.float32)
    gamma = (np.sqrt(0.1 * np.eye(c) + ops.PEDESTAL) + rng.uniform(-0.01, 0.05, (c, c))).astype(np.float32)  # some below the bound
    with torch.no_grad():
        g.beta.copy_(t(beta)); g.gamma.copy_(t(gamma))
        got = g.forward_rows(t(x))
    assert_close(n(got), ops.gdn(x, beta, gamma, inverse), what="gdn")


@pytest.mark.parametrize("rows,c,inverse", [(40_003, 128, False), (140_001, 128, True), (33_000, 192, True)])
def test_gdn_large_sets_take_the_fused_kernel(rows, c, inverse):
    """From 32 k rows on `pcc_gdn_fwd` runs `k_gdn_bf` (|x| split into bf16 planes while staging, no plane round trip):
    against the oracle, and bit for bit against the general kernel -- a row's result does not depend on how many rows the
    call has, so the first rows of the large call must equal a small call's (which takes the general kernel)."""
    from unified_point_cloud_compression_amd.model.blocks import MinkowskiGDN
    rng = np.random.default_rng(rows)
    x = (rng.standard_normal((rows, c)) * np.exp(rng.uniform(-3, 3, (rows, 1)))).astype(np.float32)
    g = MinkowskiGDN(c, inverse=inverse).to(dev()).eval()
    beta = (np.sqrt(1 + ops.PEDESTAL) + rng.uniform(0, 0.3, c)).astype(np.float32)
    gamma = (np.sqrt(0.1 * np.eye(c) + ops.PEDESTAL) + rng.uniform(-0.01, 0.05, (c, c))).astype(np.float32)
    with torch.no_grad():
        g.beta.copy_(t(beta)); g.gamma.copy_(t(gamma))
        xt = t(x)
        got = g.forward_rows(xt)
        small = g.forward_rows(xt[:1500].contiguous())
        tail = g.forward_rows(xt[rows - 1111:].contiguous())
    assert_close(n(got)[::7], ops.gdn(x[::7], beta, gamma, inverse), what="gdn, fused kernel")
    assert torch.equal(got[:1500], small) and torch.equal(got[rows - 1111:], tail)


def test_weight_offset_order_hook():
    """`sparse.WEIGHT_OFFSET_ORDER = "z_fastest"` re-indexes checkpoint weights while packing (the SURVEY A.3 hedge):
    a module fed the z-fastest enumeration of the same kernel gives the same output, in inference and under autograd."""
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    from unified_point_cloud_compression_amd import sparse as S
    keys = cloud_keys(3, 14, 0.2, 1)
    C = co.unpack_keys(keys)
    rng = np.random.default_rng(2)
    f = rng.standard_normal((len(keys), 32)).astype(np.float32)
    conv = ME.MinkowskiConvolution(32, 64, kernel_size=3, stride=1, bias=False, dimension=3).to(dev())
    x = ME.SparseTensor(features=t(f), coordinates=t(C.astype(np.int32)), device=dev())
    with torch.no_grad():
        ref = conv(x).F.clone()
    idx = np.arange(27)
    ix, iy, iz = idx % 3, (idx // 3) % 3, idx // 9
    zf = iz + 3 * iy + 9 * ix                      # position of native offset idx in a z-fastest enumeration
    w_native = conv.kernel.detach().clone()
    w_zf = torch.empty_like(w_native)
    w_zf[t(zf).long()] = w_native
    try:
        S.WEIGHT_OFFSET_ORDER = "z_fastest"
        with torch.no_grad():
            conv.kernel.copy_(w_zf)
            got = conv(x).F.clone()
        assert torch.equal(got, ref)
        xg = ME.SparseTensor(features=t(f).requires_grad_(True), coordinates=t(C.astype(np.int32)), device=dev())
        out = conv(xg).F
        assert torch.allclose(out, ref, atol=1e-5)
        out.square().sum().backward()
        g_zf = conv.kernel.grad.clone()
    finally:
        S.WEIGHT_OFFSET_ORDER = "x_fastest"
    conv.kernel.grad = None
    with torch.no_grad():
        conv.kernel.copy_(w_native)
    xg = ME.SparseTensor(features=t(f).requires_grad_(True), coordinates=t(C.astype(np.int32)), device=dev())
    conv(xg).F.square().sum().backward()
    assert torch.allclose(g_zf[t(zf).long()], conv.kernel.grad, atol=1e-4, rtol=1e-4)


def test_fallback_kernels_in_subprocess():
    """The kernels behind the size limits of the fast paths (pointer-addressed MFMA gathers for feature arrays over
    4 GB, the generic wave16 kernel, sort-based coordinate sets) are selected by load-time switches: run the feature
    tests of this file once more in a child process with the fast paths off."""
    import os
    import subprocess
    import sys
    if os.environ.get("PCC_TEST_CHILD"):
        pytest.skip("already the child")
    env = dict(os.environ, PCC_TEST_CHILD="1", PCC_MFMA_BUF="0", PCC_WAVE16_ZRUN="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_map_conv.py", "-x", "-q", "-k",
                        "features or strided or pair_list or gdn or row_tails"], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]


def test_unsupported_shape_fails_loudly():
    from unified_point_cloud_compression_amd import sparse as S, lib as L
    with pytest.raises(L.PccError):
        S.PackedConv().get(torch.nn.Parameter(torch.zeros(27, 24, 24, device=dev())))
    with pytest.raises(L.PccError):
        L.ptr(torch.zeros(4))      # CPU tensors never reach the library


@pytest.mark.parametrize("cin,cout,ks", [(128, 128, 3), (192, 256, 3), (128, 64, 5), (32, 32, 3)])
def test_split_path_accuracy(cin, cout, ks):
    """The default MFMA path evaluates fp32 products on the bf16 matrix pipe from an exact 3-way bf16 split of both
    operands (six cross terms, fp32 accumulation).  Its error against a float64 evaluation must be at the level of the
    fp32-input MFMA path's own rounding error -- fp32 accuracy, not bf16 accuracy -- and both must meet the 1e-4 bar."""
    from unified_point_cloud_compression_amd import sparse as S, lib as L
    keys = cloud_keys(31, 26, 0.15, 1)
    cs = _cs(keys, 1)
    rng = np.random.default_rng(cin * 7 + cout)
    f = (rng.standard_normal((len(keys), cin)) * np.exp(rng.uniform(-3, 3, (len(keys), cin)))).astype(np.float32)   # 3 decades of magnitudes
    K = ks ** 3
    W = (rng.standard_normal((K, cin, cout)) / np.sqrt(cin * 10)).astype(np.float32)
    b = rng.standard_normal((1, cout)).astype(np.float32)
    m = cs.kernel_map(cs, ks)
    nbr = co.kernel_map(keys, keys, ks, 1)
    want64 = np.zeros((len(keys), cout)) + b.astype(np.float64)
    f64, W64 = f.astype(np.float64), W.astype(np.float64)
    for k in range(K):
        o = np.nonzero(nbr[k] >= 0)[0]
        want64[o] += f64[nbr[k, o]] @ W64[k]
    scale = np.abs(want64).max()
    pk = S.PackedConv().get(torch.nn.Parameter(t(W)))
    old_pair = S.PAIR_MIN_K
    S.PAIR_MIN_K = 1 << 30
    try:
        with L.arith_scope(L.ARITH_BF6):                              # (explicit: the environment may have selected another form)
            got_split = n(S.conv_forward(t(f), pk, t(b), K, cin, cout, m, cs.n))
        with L.arith_scope(L.ARITH_F32):
            got_fp32 = n(S.conv_forward(t(f), pk, t(b), K, cin, cout, m, cs.n))
    finally:
        S.PAIR_MIN_K = old_pair
    e_split = np.abs(got_split - want64).max() / scale
    e_fp32 = np.abs(got_fp32 - want64).max() / scale
    rms_split = np.sqrt(((got_split - want64) ** 2).mean()) / scale
    rms_fp32 = np.sqrt(((got_fp32 - want64) ** 2).mean()) / scale
    print(f"cin={cin} cout={cout} K={K}: max err / scale split {e_split:.2e} fp32 {e_fp32:.2e}; rms split {rms_split:.2e} fp32 {rms_fp32:.2e}")
    assert e_fp32 < 5e-6 and e_split < 5e-6                      # both far inside the 1e-4 bar
    assert rms_split <= 3.0 * rms_fp32 + 1e-8                    # fp32-level error, not bf16-level (which would be ~1e-3)
    assert_close(got_split, want64, what="split path vs float64")
    assert not np.array_equal(got_split, got_fp32) or cin < 32   # the two paths really are different kernels



@pytest.mark.parametrize("spread", ["normal", "rows", "elements", "weights"])
def test_dense_products_accuracy(spread):
    """Dense products of the generative transposed convolutions in scaled fp16 pairs (three MFMA terms, `k_gemm_h2`): error
    against a float64 evaluation at the level of the fp32 accumulation itself -- not above the six-term bf16 form's by more
    than 3x, and below 4e-6 of the row's largest output -- for plain data and for rows / elements / weights whose
    magnitudes spread over e^+-12 (the scales are per feature row and per weight column)."""
    from unified_point_cloud_compression_amd import sparse as S, lib as L
    rng = np.random.default_rng(5)
    n_rows, cin, ncol = 4096 + 77, 128, 2048 + 64                              # 33 row tiles (the last one ragged), 17 column blocks (the last one half)
    x = rng.standard_normal((n_rows, cin)).astype(np.float32)
    w = (rng.standard_normal((cin, ncol)) / np.sqrt(cin)).astype(np.float32)
    if spread == "rows":
        x = (np.maximum(x, 0) * np.exp(rng.standard_normal((n_rows, 1)) * 4)).astype(np.float32)
        x[5] = 0.0                                                           # an all-zero row
    elif spread == "elements":
        x = (x * np.exp(rng.standard_normal(x.shape) * 4)).astype(np.float32)
    elif spread == "weights":
        w = (w * np.exp(rng.standard_normal(w.shape) * 3)).astype(np.float32)
        w[:, 7] = 0.0
    want = x.astype(np.float64) @ w.astype(np.float64)
    K, cout = 8, ncol // 8                                                     # [cin, K*cout] flat operand of a k2-s2 transposed conv
    W = torch.nn.Parameter(t(np.ascontiguousarray(w.reshape(cin, K, cout).transpose(1, 0, 2))))
    first = torch.arange(0, n_rows * K + 1, dtype=torch.int32, device=dev())   # one pair per output row: T itself comes back
    pair_ids = torch.arange(0, n_rows * K, dtype=torch.int32, device=dev())
    res = {}
    for name, form in (("h", L.ARITH_H3), ("bf", L.ARITH_BF6)):
        with L.arith_scope(form):
            pk = S.PackedConv(True).get(W)
            got = S.convt_forward_csr(t(x), pk, None, K, cin, cout, (first, pair_ids), n_rows * K)
        res[name] = n(got).reshape(n_rows, ncol).astype(np.float64)
    scale = np.abs(want).max(1, keepdims=True) + 1e-300
    err = {k: (np.abs(v - want) / scale).max() for k, v in res.items()}
    assert err["h"] <= 4e-6, err
    assert err["h"] <= 3 * err["bf"] + 1e-7, err


def _wide_product(rows=6000 + 77, cin=128, K=343, cout=16, seed=9):
    """T = X Wflat of a composite level's shape (7x7x7 offsets x 16 channels = 5 488 columns: 21.5 column blocks of 256, the
    last one half) through `pcc_convt_fwd_csr` with one pair per output row, so that the product buffer itself comes back."""
    from unified_point_cloud_compression_amd import sparse as S
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, cin)) * np.exp(rng.standard_normal((rows, 1)))).astype(np.float32)
    w = (rng.standard_normal((cin, K * cout)) / np.sqrt(cin)).astype(np.float32)
    W = torch.nn.Parameter(t(np.ascontiguousarray(w.reshape(cin, K, cout).transpose(1, 0, 2))))
    first = torch.arange(0, rows * K + 1, dtype=torch.int32, device=dev())
    pair_ids = torch.arange(0, rows * K, dtype=torch.int32, device=dev())
    pk = S.PackedConv(True).get(W)
    got = S.convt_forward_csr(t(x), pk, None, K, cin, cout, (first, pair_ids), rows * K)
    return x, w, n(got).reshape(rows, K * cout)


def _wide_product_sha():
    import hashlib
    return hashlib.sha256(_wide_product()[2].tobytes()).hexdigest()


def test_wide_tile_dense_products_equal_the_128_wide_tile_bit_for_bit():
    """Round 4: a 128 x 256 tile of the dense products (`k_gemm_h2<NCH, 4>`, measured 2 % slower than the 128 x 128 tile and
    therefore off: DESIGN.md).  Same chunk order, same term order per element, so the two must agree BIT FOR BIT (a child
    process runs the wide tile: `PCC_GEMM_WIDE` is a load-time switch); ragged last row tile, half-empty last column block;
    and against float64."""
    import hashlib
    import os
    import subprocess
    import sys
    x, w, got = _wide_product()
    want = x.astype(np.float64) @ w.astype(np.float64)
    scale = np.abs(want).max(1, keepdims=True) + 1e-300
    assert (np.abs(got - want) / scale).max() <= 4e-6
    if os.environ.get("PCC_GEMM_WIDE", "0") != "0":
        pytest.skip("this process already runs the wide tile")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "from tests.test_gpu_map_conv import _wide_product_sha; print('SHA', _wide_product_sha())"],
                       cwd=root, env=dict(os.environ, PCC_GEMM_WIDE="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    wide = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("SHA ")][-1]
    assert wide == hashlib.sha256(got.tobytes()).hexdigest()


def test_pair_list_products_accuracy_with_spread_rows():
    """5x5x5 convolution in the pair-list form on scaled fp16 pairs (`k_pair_h2`): rows whose magnitudes spread over
    e^+-12 -- the scale is per input row and per (offset, column) of the weights -- against float64 and against the
    six-term bf16 form (`PCC_ARITH_BF6`)."""
    from unified_point_cloud_compression_amd import sparse as S, lib as L
    rng = np.random.default_rng(17)
    keys = cloud_keys(5, 40, 0.08, 1)
    cs = _cs(keys, 1)
    cin = cout = 128
    K = 125
    f = (np.maximum(rng.standard_normal((len(keys), cin)), 0) * np.exp(rng.standard_normal((len(keys), 1)) * 4)).astype(np.float32)
    W = (rng.standard_normal((K, cin, cout)) / np.sqrt(cin * 8) * np.exp(rng.standard_normal((K, 1, cout)))).astype(np.float32)
    m = cs.kernel_map(cs, 5)
    nbr = co.kernel_map(keys, keys, 5, 1)
    want = np.zeros((len(keys), cout))
    f64, W64 = f.astype(np.float64), W.astype(np.float64)
    mag = np.zeros((len(keys), 1))                                             # largest input-row magnitude feeding an output row
    rowmax = np.abs(f64).max(1)
    for k in range(K):
        o = np.nonzero(nbr[k] >= 0)[0]
        want[o] += f64[nbr[k, o]] @ W64[k]
        mag[o, 0] = np.maximum(mag[o, 0], rowmax[nbr[k, o]])
    res = {}
    for name, form in (("h", L.ARITH_H3), ("bf", L.ARITH_BF6)):
        with L.arith_scope(form):
            pk = S.PackedConv().get(torch.nn.Parameter(t(W)))
            assert m.pair_plan() is not None                                   # sparse 5x5x5 map: the pair-list form runs
            res[name] = n(S.conv_forward(t(f), pk, None, K, cin, cout, m, cs.n)).astype(np.float64)
    scale = mag * np.abs(W64).max() * np.sqrt(cin) + 1e-300                    # size of one neighbour's contribution
    err = {k: (np.abs(v - want) / scale).max() for k, v in res.items()}
    assert err["h"] <= 2e-5 and err["h"] <= 4 * err["bf"] + 1e-7, err
    import os
    if not os.environ.get("PCC_TEST_CHILD"):                                   # (the child run switches the fast paths off)
        assert not np.array_equal(res["h"], res["bf"])                         # the two forms really are different kernels
