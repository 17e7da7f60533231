"""Occupancy head in one pass (`predict_i`, reference `model/transforms.py:141-160`): conv k3 -> ReLU -> conv k3 -> logit
with the hidden features kept on chip (`pcc_conv_head_fwd`), and the band-ordered tile table its stencil kernel walks
(`pcc_band_tiles_build`), against the oracle."""
import numpy as np
import pytest
import torch

from oracle import coords as co, ops
from tests.util import dev, t, n, cloud_keys, assert_close

pytestmark = pytest.mark.gpu


def _cs(keys, ts):
    from unified_point_cloud_compression_amd import sparse as S
    C = co.unpack_keys(keys)
    return S.CoordSet(t(keys), len(keys), ts, S.Bounds(0, C[:, 1:].min(0), C[:, 1:].max(0)))


@pytest.fixture()
def small_bands():
    from unified_point_cloud_compression_amd import sparse as S
    old = S.BAND_MIN_ROWS, S.BAND_COUNT
    S.BAND_MIN_ROWS, S.BAND_COUNT = 1, 5
    yield
    S.BAND_MIN_ROWS, S.BAND_COUNT = old


@pytest.mark.parametrize("ts,shift", [(1, 0), (2, -6)])
def test_band_tiles_partition_the_rows(ts, shift, small_bands):
    """Every row in exactly one tile; a tile = <= 16 consecutive rows of one (y-band, x) run; tiles are numbered band-major
    with x ascending inside a band."""
    keys = cloud_keys(7, 37, 0.2, ts)
    C = co.unpack_keys(keys)
    C[:, 1:] += shift * ts                                       # negative coordinates too
    keys = np.unique(co.pack_keys(C))
    C = co.unpack_keys(keys)
    cs = _cs(keys, ts)
    tiles, n_tiles = cs.band_tiles()
    nt = int(n_tiles.item())
    tw = n(tiles)[:nt].astype(np.int64)
    row0, cnt = tw & 0x07FFFFFF, ((tw >> 27) & 31) + 1
    assert cnt.max() <= 16 and cnt.min() >= 1
    cover = np.zeros(len(keys), np.int32)
    ny = (C[:, 2].max() - C[:, 2].min()) // ts + 1
    band_h = -(-ny // 5)
    band_of = ((C[:, 2] - C[:, 2].min()) // ts) // band_h
    order = []
    for r0, c in zip(row0, cnt):
        cover[r0:r0 + c] += 1
        assert len(set(C[r0:r0 + c, 1])) == 1 and len(set(band_of[r0:r0 + c])) == 1
        order.append((band_of[r0], C[r0, 1], r0))
    assert np.all(cover == 1)
    assert order == sorted(order)


@pytest.mark.parametrize("cin,cmid,bands", [(32, 16, True), (32, 16, False), (16, 8, True), (32, 8, False)])
def test_fused_head_matches_oracle(cin, cmid, bands, small_bands):
    from unified_point_cloud_compression_amd import sparse as S
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    from unified_point_cloud_compression_amd.MinkowskiEngine.sparse_tensor import SparseTensor
    from unified_point_cloud_compression_amd.model.transforms import SparseSynthesisTransform as G
    rng = np.random.default_rng(cin + cmid)
    keys = cloud_keys(11, 30, 0.25, 1)
    keys = keys[: len(keys) - (len(keys) % 16) + 5]              # a ragged last tile
    cs = _cs(keys, 1)
    x = rng.standard_normal((len(keys), cin)).astype(np.float32)
    c0 = ME.MinkowskiConvolution(cin, cmid, kernel_size=3, stride=1, bias=True, dimension=3).to(dev())
    c2 = ME.MinkowskiConvolution(cmid, 1, kernel_size=3, stride=1, bias=True, dimension=3).to(dev())
    with torch.no_grad():
        c0.kernel.mul_(3.0)
        c2.kernel.mul_(3.0)
    head = torch.nn.Sequential(c0, ME.MinkowskiReLU(), c2)
    nbr = co.kernel_map(keys, keys, 3, 1)
    h = ops.relu(ops.conv(x, n(c0.kernel), n(c0.bias), nbr))
    want = ops.conv(h, n(c2.kernel), n(c2.bias), nbr)
    old = S.BAND_TILES, S.HEAD_FUSED
    try:
        S.BAND_TILES, S.HEAD_FUSED = bands, True
        assert (cs.band_tiles() is not None) == bands
        with torch.no_grad():
            xt = SparseTensor._from_canonical(cs, t(x))
            got = G._predict(head, xt)
            S.HEAD_FUSED = False
            ref = G._predict(head, xt)
            seq = head(xt)                                        # plain module calls
    finally:
        S.BAND_TILES, S.HEAD_FUSED = old
    assert_close(n(got.F), want, what="fused head vs oracle")
    assert_close(n(ref.F), want, what="layer-wise head vs oracle")
    assert_close(n(seq.F), want, what="Sequential head vs oracle")
    assert float((got.F - ref.F).abs().max().item()) <= 1e-5
    # deterministic: bit-identical on a second evaluation
    with torch.no_grad():
        S.BAND_TILES = bands
        try:
            again = G._predict(head, SparseTensor._from_canonical(cs, t(x)))
        finally:
            S.BAND_TILES = old[0]
    assert torch.equal(again.F, got.F)


def _two_batch_keys(seed, ts, shift):
    C = []
    for b, (size, p) in enumerate(((21, 0.3), (17, 0.2))):
        c = co.unpack_keys(cloud_keys(seed + b, size, p, ts))
        c[:, 0] = b
        c[:, 1:] += shift * ts
        C.append(c)
    return np.unique(co.pack_keys(np.concatenate(C)))


@pytest.mark.parametrize("ts,shift,cin,cout", [(1, 0, 16, 1), (2, -5, 16, 1), (2, -5, 8, 3), (1, -3, 32, 4), (1, 0, 64, 1), (2, -5, 32, 1)])
def test_thin_conv_from_grid_matches_oracle(ts, shift, cin, cout):
    """`pcc_conv_thin_grid_fwd`: 3x3x3 conv to <= 4 channels, neighbour rows from the bitmap + rank (no kernel map);
    two batch entries, negative coordinates, rows on every face of the bounding lattice.  (cin 32 / 64 with one output
    channel: the projections run on the matrix pipe, `k_thin_project_mfma`.)"""
    from unified_point_cloud_compression_amd import sparse as S
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    rng = np.random.default_rng(cin * 7 + cout)
    keys = _two_batch_keys(3, ts, shift)
    C = co.unpack_keys(keys)
    cs = S.CoordSet(t(keys), len(keys), ts, S.Bounds(1, C[:, 1:].min(0), C[:, 1:].max(0)))
    x = rng.standard_normal((len(keys), cin)).astype(np.float32)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=3, stride=1, bias=True, dimension=3).to(dev())
    with torch.no_grad():
        conv.kernel.mul_(3.0)
        conv.bias.add_(0.5)
        w = conv._packed.get(conv.kernel, state_dict_order=True)
        got = S.conv_thin_grid_forward(t(x), w, conv.bias, cin, cout, cs)
        again = S.conv_thin_grid_forward(t(x), w, conv.bias, cin, cout, cs)
    want = ops.conv(x, n(conv.kernel), n(conv.bias), co.kernel_map(keys, keys, 3, ts))
    assert_close(n(got), want, what="thin conv from grid vs oracle")
    assert torch.equal(got, again)
    if cin == 16 and cout == 1:        # the z-folded planes (large sets by default): same result up to the summation order
        from unified_point_cloud_compression_amd import lib as L
        try:
            L.call("pcc_set_thin_z_min_rows", 0)
            with torch.no_grad():
                zf = S.conv_thin_grid_forward(t(x), w, conv.bias, cin, cout, cs)
        finally:
            L.call("pcc_set_thin_z_min_rows", 1 << 20)
        assert_close(n(zf), want, what="thin conv, z-folded planes vs oracle")
        assert_close(n(zf), n(got), atol=5e-6, rtol=1e-6, what="z-folded planes vs 27-plane form")


@pytest.mark.parametrize("ts_in,shift,cin,cout", [(2, 0, 16, 8), (4, -3, 16, 8), (2, -3, 32, 16), (4, 0, 32, 64), (2, 0, 32, 32)])
def test_composite_gather_presence_from_grid_equals_table_path(ts_in, shift, cin, cout):
    """`pcc_convt_fwd_csr_grid` == `pcc_convt_fwd_csr` with the 3x3x3 neighbour table, bit for bit (same pair order, same
    presence flags), on a two-batch input.  8 channels: two lanes per row (loop form of the presence probe); 16 / 32 / 64:
    4 / 8 / 16 lanes per row (branch-free windows, 3 / 2 / 1 columns per lane)."""
    from unified_point_cloud_compression_amd import sparse as S, lib as L
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    rng = np.random.default_rng(ts_in)
    keys = _two_batch_keys(9, ts_in, shift)
    C = co.unpack_keys(keys)
    cs = S.CoordSet(t(keys), len(keys), ts_in, S.Bounds(1, C[:, 1:].min(0), C[:, 1:].max(0)))
    ts_out = ts_in // 2
    out_set = cs.expand(5, ts_out, want_csr=False)
    assert out_set.grid() is not None
    csr7 = cs.csr_for(out_set.keys, out_set.n, 7, ts_out)
    x = t(rng.standard_normal((len(keys), cin)).astype(np.float32))
    gen = ME.MinkowskiGenerativeConvolutionTranspose(cin, cout, kernel_size=7, stride=2, bias=True, dimension=3).to(dev())
    ex_bias = t(rng.standard_normal((27, cout)).astype(np.float32))
    with torch.no_grad():
        w = gen._packed.get(gen.kernel)
        a = S.convt_forward_csr_grid(x, w, gen.bias, 343, cin, cout, csr7, out_set, L.ACT_RELU, ex_bias)
        kmap3 = out_set.kernel_map(out_set, 3)
        b = S.convt_forward_csr(x, w, gen.bias, 343, cin, cout, csr7, out_set.n, act=L.ACT_RELU, ex_map=kmap3, ex_bias=ex_bias)
    assert torch.equal(a, b)
    assert float(a.abs().max().item()) > 0


@pytest.mark.parametrize("ts_in,shift,with_ex", [(2, 0, True), (4, -3, True), (2, -2, False)])
def test_chunked_composite_equals_one_pass_bit_for_bit(ts_in, shift, with_ex):
    """`pcc_convt_fwd_csr_chunked` (products staged in cache-sized parent chunks, partial sums carried in the output) ==
    the one-pass form, bit for bit: several chunks, a chunk that spans the batch boundary, output rows with an empty pair
    list (owned by exactly one chunk), with and without the per-neighbour constant."""
    from unified_point_cloud_compression_amd import sparse as S, lib as L
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    rng = np.random.default_rng(ts_in + 10)
    keys = _two_batch_keys(9, ts_in, shift)
    assert len(keys) > 3 * 1024                                             # >= 4 chunks of 1024 parents
    C = co.unpack_keys(keys)
    cs = S.CoordSet(t(keys), len(keys), ts_in, S.Bounds(1, C[:, 1:].min(0), C[:, 1:].max(0)))
    ts_out = ts_in // 2
    full = cs.expand(5, ts_out, want_csr=False)
    ok = n(full.keys)[:full.n]
    Co = co.unpack_keys(ok)
    lonely = np.array([[0, Co[:, 1].max() + 40 * ts_out, 0, 0], [1, Co[:, 1].max() + 40 * ts_out, 2 * ts_out, 0],
                       [0, Co[:, 1].min() - 20 * ts_out, 0, 0]], np.int64)   # no parent within reach: empty pair lists
    okeys = np.unique(np.concatenate([ok, co.pack_keys(lonely)]))
    Co = co.unpack_keys(okeys)
    out_set = S.CoordSet(t(okeys), len(okeys), ts_out, S.Bounds(1, Co[:, 1:].min(0), Co[:, 1:].max(0)))
    csr7 = cs.csr_for(out_set.keys, out_set.n, 7, ts_out)
    first = n(csr7[0])
    assert (np.diff(first[:out_set.n + 1]) == 0).sum() == 3
    cin, cout = 32, 16
    x = t(rng.standard_normal((len(keys), cin)).astype(np.float32))
    gen = ME.MinkowskiGenerativeConvolutionTranspose(cin, cout, kernel_size=7, stride=2, bias=True, dimension=3).to(dev())
    ex_bias = t(rng.standard_normal((27, cout)).astype(np.float32)) if with_ex else None
    lib = L.load()
    try:
        L.call("pcc_set_t_chunk_bytes", 1 << 20)                             # -> the 1024-row minimum: 4 chunks
        with torch.no_grad():
            w = gen._packed.get(gen.kernel)
            a = S.convt_forward_csr_chunked(x, w, gen.bias, 343, cin, cout, csr7, cs, out_set, L.ACT_RELU, ex_bias)
            a2 = S.convt_forward_csr_chunked(x, w, gen.bias, 343, cin, cout, csr7, cs, out_set, L.ACT_RELU, ex_bias)
            if with_ex:
                b = S.convt_forward_csr_grid(x, w, gen.bias, 343, cin, cout, csr7, out_set, L.ACT_RELU, ex_bias)
            else:
                b = S.convt_forward_csr(x, w, gen.bias, 343, cin, cout, csr7, out_set.n, act=L.ACT_RELU)
    finally:
        L.call("pcc_set_t_chunk_bytes", 96 << 20)
    assert lib.pcc_convt_chunk_t_bytes(len(keys), 343, cout) >= 1024 * 343 * cout * 4
    assert torch.equal(a, b) and torch.equal(a, a2)
    assert float(a.abs().max().item()) > 0


def test_pair_lists_with_z_fastest_offset_numbering():
    """`pcc_coords_expand_grid_csr_zk` = the lists of `pcc_coords_expand_grid_csr` with every pair's kernel offset renumbered
    iz + 7 iy + 49 ix (same rows, same order inside a row)."""
    from unified_point_cloud_compression_amd import sparse as S
    keys = _two_batch_keys(4, 2, -3)
    C = co.unpack_keys(keys)
    cs = S.CoordSet(t(keys), len(keys), 2, S.Bounds(1, C[:, 1:].min(0), C[:, 1:].max(0)))
    out_set = cs.expand(5, 1, want_csr=False)
    fa, pa = cs.csr_for(out_set.keys, out_set.n, 7, 1)
    fb, pb = cs.csr_for(out_set.keys, out_set.n, 7, 1, zk=True)
    fa, fb = n(fa), n(fb)
    assert np.array_equal(fa, fb)
    P = int(fa[out_set.n])
    pa, pb = n(pa)[:P].astype(np.int64), n(pb)[:P].astype(np.int64)
    row_a, k_a = pa // 343, pa % 343
    ix, iy, iz = k_a % 7, (k_a // 7) % 7, k_a // 49
    assert np.array_equal(pb // 343, row_a) and np.array_equal(pb % 343, iz + 7 * iy + 49 * ix)
    assert P > out_set.n                                   # several parents per child on average


@pytest.mark.parametrize("ks,ts_in,shift,zk,dense", [(7, 2, -3, True, False), (7, 4, 0, False, False), (5, 2, -3, False, False), (7, 2, 0, True, True)])
def test_one_pass_slotted_pair_lists_equal_the_three_launch_form(ks, ts_in, shift, zk, dense):
    """Round 4: `pcc_coords_expand_grid_csr_slots` (probe once, positions from a scan INSIDE the workgroup, lists in
    per-workgroup slots) holds exactly the lists of the count / scan / fill form -- same pairs, same order inside a row -- and the
    composite gather-sum over them gives the same bits.  Two batch entries, negative coordinates, a row count that is not a
    multiple of 256, and a DENSE block whose rows carry up to 64 pairs (more than the 16 parked in LDS: the re-probe path, and
    workgroup totals beyond the LDS stage: the write-through path)."""
    from unified_point_cloud_compression_amd import sparse as S, lib as L
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    if dense:
        g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(12), indexing="ij"), -1).reshape(-1, 3) * ts_in
        keys = np.unique(co.pack_keys(np.concatenate([np.zeros((len(g), 1), np.int64), g], 1)))
    else:
        keys = _two_batch_keys(9, ts_in, shift)
    C = co.unpack_keys(keys)
    cs = S.CoordSet(t(keys), len(keys), ts_in, S.Bounds(int(C[:, 0].max()), C[:, 1:].min(0), C[:, 1:].max(0)))
    ts_out = ts_in // 2
    out_set = cs.expand(5, ts_out, want_csr=False)
    n_out = out_set.n
    first, pids = cs.csr_for(out_set.keys, n_out, ks, ts_out, zk=zk)
    total = L.counter()
    sl = cs.csr_for(out_set.keys, n_out, ks, ts_out, zk=zk, slots=True, total=total)
    assert len(sl) == 3
    f, p = n(first).astype(np.int64), n(pids)
    sf, sp, we = n(sl[0]).astype(np.int64), n(sl[1]), n(sl[2]).astype(np.int64)
    assert S.csr_pair_total(sl, n_out) == int(f[n_out]) == int(L.read(total)[0])
    ends = np.empty(n_out, np.int64)
    ends[:-1] = sf[1:]
    last = np.arange(n_out) % 256 == 255
    last[-1] = True
    ends[last] = we[np.arange(n_out)[last] // 256]
    assert np.array_equal(ends - sf, np.diff(f[:n_out + 1]))                  # same list lengths, row by row
    if dense:
        assert (ends - sf).max() > 16
    rows = np.random.default_rng(0).permutation(n_out)[:4000]
    for o in rows:
        assert np.array_equal(sp[sf[o]:ends[o]], p[f[o]:f[o + 1]]), o
    if ks == 7:                                                                # ... and the composite level over them: same bits
        rng = np.random.default_rng(ks + ts_in)
        cin, cout = 32, 16
        x = t(rng.standard_normal((len(keys), cin)).astype(np.float32))
        gen = ME.MinkowskiGenerativeConvolutionTranspose(cin, cout, kernel_size=7, stride=2, bias=True, dimension=3).to(dev())
        ex_bias = t(rng.standard_normal((27, cout)).astype(np.float32))
        with torch.no_grad():
            w = gen._packed.get(gen.kernel)
            a = S.convt_forward_csr_grid(x, w, gen.bias, 343, cin, cout, (first, pids), out_set, L.ACT_RELU, ex_bias)
            b = S.convt_forward_csr_grid(x, w, gen.bias, 343, cin, cout, sl, out_set, L.ACT_RELU, ex_bias)
        assert torch.equal(a, b) and float(a.abs().max().item()) > 0
