"""End-to-end parity of UnifiedModel.compress/decompress (HIP path) against the oracle and the committed goldens."""
import numpy as np
import pytest
import torch

from oracle import codec, coords as co, ops
from tests.util import dev, t, n, load_params, assert_close
from tests.golden import make_golden

pytestmark = pytest.mark.gpu


def _model(cfg, P, coder="symbols"):
    """coder="symbols": int32 symbol tensors cross the entropy-coder boundary (symbol-level parity checks);
    "pcc_streams" / "ans": real rANS byte strings."""
    import copy
    from unified_point_cloud_compression_amd.model import UnifiedModel
    cfg = copy.deepcopy(cfg)
    cfg["entropy_model"]["entropy_coder"] = coder
    m = load_params(UnifiedModel(cfg), P).to(dev()).eval()
    m.update()
    return m


def _sym_close(got, want, what):
    """Integer symbols: equal except where the pre-rounding value sat within float noise of .5 (must be rare, +-1)."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, what
    diff = got != want
    assert diff.mean() <= 2e-3, f"{what}: {diff.mean():.4%} symbols differ"
    assert np.abs(got[diff] - want[diff]).max(initial=0) <= 1, what


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_compress_decompress_matches_oracle_and_golden(seed):
    from unified_point_cloud_compression_amd import synth
    adaptive = seed == 2
    cfg = codec.small_config(adaptive=adaptive, offsets=adaptive)
    P = codec.random_params(cfg, seed, gain=make_golden.GAIN[seed])
    model = _model(cfg, P)
    pc = synth.random_block(seed, 32, 0.08)
    q = np.array([[0.3 + 0.2 * seed, 0.6]], dtype=np.float32)
    gold = np.load(f"tests/golden/codec_seed{seed}.npz")

    streams, shapes, ks, coords, qs = model.compress(t(pc), t(q), block_size=1024)
    assert len(streams) == 1 and ks[0] == gold["k"].tolist() and shapes[0] == [len(gold["z_keys"])]
    y_keys = n(coords[0]._pcc_cset.keys)[:coords[0].shape[0]]
    assert np.array_equal(y_keys, gold["y_keys"])                              # coordinates bit exact
    assert np.array_equal(co.pack_keys(n(coords[0])), gold["y_keys"])
    y_sym, z_sym = n(streams[0][0]), n(streams[0][1])
    _sym_close(y_sym, gold["y_symbols"], "y symbols")
    _sym_close(z_sym, gold["z_symbols"], "z symbols")

    # decode the GPU's own symbols on both sides: identical integer inputs, so everything downstream must agree
    trace_g, trace_o = {}, {}
    rec = model.decompress(coordinates=coords, strings=streams, shape=shapes, k=ks, q_vals=qs, trace=trace_g)
    blk = dict(y_keys=y_keys, y_symbols=y_sym, z_symbols=z_sym, k=ks[0], q=q)
    rec_o = codec.decompress(P, cfg, [blk], trace=trace_o)
    for lvl in range(3):
        assert np.array_equal(n(trace_g[f"keys_{lvl}"]), trace_o[f"keys_{lvl}"]), f"generative coords level {lvl}"
        assert_close(n(trace_g[f"feats_{lvl}"]), trace_o[f"feats_{lvl}"], what=f"features level {lvl}")
        assert_close(n(trace_g[f"logit_{lvl}"]), trace_o[f"logit_{lvl}"], what=f"logits level {lvl}")
        # the mask is bit exact for the logits the GPU actually produced ...
        gl = n(trace_g[f"logit_{lvl}"])[:, 0]
        assert np.array_equal(n(trace_g[f"mask_{lvl}"]), ops.topk_mask(gl, ks[0][lvl]))
        # ... and equals the oracle's own mask (no near-ties at the k-th logit for these seeds)
        assert np.array_equal(n(trace_g[f"mask_{lvl}"]), trace_o[f"mask_{lvl}"]), f"occupancy mask level {lvl}"
    rec = n(rec)
    assert rec.shape == rec_o.shape == (gold["n_points"], 6)
    assert np.array_equal(rec[:, :3], rec_o[:, :3])
    # colours are rounded to 8 bit: allow one level where the pre-rounding value sat on a boundary
    lv, lv_o = np.rint(rec[:, 3:] * 255).astype(int), np.rint(rec_o[:, 3:] * 255).astype(int)
    assert np.abs(lv - lv_o).max() <= 1
    assert (lv != lv_o).mean() < 5e-3
    if np.array_equal(y_sym, gold["y_symbols"]) and np.array_equal(z_sym, gold["z_symbols"]):
        assert np.array_equal(rec[:, :3], gold["recon"][:, :3])

    # encoder / decoder agreement on the HIP path itself: decode(encode(x)) reproduces the input geometry count
    assert rec.shape[0] == ks[0][2][0]
    # rate from likelihoods (`loss.py:77-79`) within 0.5 % of the oracle's
    x = model.block_input(t(pc))
    y, _ = model.g_a(x)
    y_lik, z_lik = model.entropy_model.likelihoods(y, t(q))
    bits = float(-(torch.log2(y_lik.double()).sum() + torch.log2(z_lik.double()).sum()))
    assert abs(bits - float(gold["bits"])) / float(gold["bits"]) < 5e-3


@pytest.mark.parametrize("bits", [6, 7, 8])
def test_r2_architecture_parity_small_surface(bits):
    """The full-width R2 architecture (128 / 192 channels: MFMA kernels, pair-list 5x5x5 convolutions, the z-run
    32->16 kernel, thin heads) end to end against the oracle on a small synthetic surface."""
    from unified_point_cloud_compression_amd import synth
    cfg = codec.R2_CONFIG
    P = codec.random_params(cfg, 0, gain=3.0)
    model = _model(cfg, P)
    pc = synth.surface_cloud(0, bits)
    q = np.array([[0.5, 0.5]], dtype=np.float32)
    streams, shapes, ks, coords, qs = model.compress(t(pc), t(q), block_size=1024)
    blocks = codec.compress(P, cfg, pc, q)
    assert len(streams) == len(blocks) == 1 and ks[0] == blocks[0]["k"]
    y_keys = n(coords[0]._pcc_cset.keys)[:coords[0].shape[0]]
    assert np.array_equal(y_keys, blocks[0]["y_keys"])
    y_sym, z_sym = n(streams[0][0]), n(streams[0][1])
    _sym_close(y_sym, blocks[0]["y_symbols"], "y symbols")
    _sym_close(z_sym, blocks[0]["z_symbols"], "z symbols")
    trace_g, trace_o = {}, {}
    rec = n(model.decompress(coordinates=coords, strings=streams, shape=shapes, k=ks, q_vals=qs, trace=trace_g))
    blk = dict(y_keys=y_keys, y_symbols=y_sym, z_symbols=z_sym, k=ks[0], q=q)
    rec_o = codec.decompress(P, cfg, [blk], trace=trace_o)
    same_sets = True
    for lvl in range(3):
        if not same_sets:
            break               # a flipped near-tie at one level changes the sets of the next: nothing left to compare
        assert np.array_equal(n(trace_g[f"keys_{lvl}"]), trace_o[f"keys_{lvl}"]), f"generative coords level {lvl}"
        assert_close(n(trace_g[f"feats_{lvl}"]), trace_o[f"feats_{lvl}"], what=f"features level {lvl}")
        assert_close(n(trace_g[f"logit_{lvl}"]), trace_o[f"logit_{lvl}"], atol=2e-4, what=f"logits level {lvl}")
        gl = n(trace_g[f"logit_{lvl}"])[:, 0]
        mg = n(trace_g[f"mask_{lvl}"])
        assert np.array_equal(mg, ops.topk_mask(gl, ks[0][lvl]))               # exact for the GPU's own logits
        mo = trace_o[f"mask_{lvl}"]
        flips = int((mg != mo).sum())
        if flips:               # only rows whose logit sits within float noise of the k-th may differ
            kth = np.sort(gl)[::-1][ks[0][lvl][0] - 1]
            assert flips <= 4 and np.all(np.abs(gl[mg != mo] - kth) < 2e-4), (lvl, flips)
            same_sets = False
    if same_sets:
        assert rec.shape == rec_o.shape and np.array_equal(rec[:, :3], rec_o[:, :3])
    assert rec.shape[0] == pc.shape[0]


@pytest.mark.parametrize("which", ["small", "r2"])
def test_fused_up_predict_matches_layerwise(which):
    """Inference evaluates an up-sampling block and its occupancy head as one composite 7x7x7 generative convolution
    (and the up-sampled features only for the kept rows): same logits up to float noise, same occupancy decisions, same
    reconstruction as the layer-by-layer path, and as the oracle."""
    from unified_point_cloud_compression_amd import synth
    from unified_point_cloud_compression_amd.model.transforms import SparseSynthesisTransform as G
    from unified_point_cloud_compression_amd.MinkowskiEngine.sparse_tensor import SparseTensor
    if which == "small":
        cfg, pc, gain = codec.small_config(), synth.random_block(4, 40, 0.08), 4.0
    else:
        cfg, pc, gain = codec.R2_CONFIG, synth.surface_cloud(0, 7), 3.0
    P = codec.random_params(cfg, 3, gain=gain)
    model = _model(cfg, P)
    q = np.array([[0.5, 0.5]], dtype=np.float32)
    streams, shapes, ks, coords, qs = model.compress(t(pc), t(q), block_size=1024)
    old_fuse, old_min, old_ratio = G.FUSE_UP_PREDICT, G.FUSE_MIN_HEAD_CHANNELS, G.FUSE_NARROW_MIN_RATIO
    try:
        G.FUSE_MIN_HEAD_CHANNELS, G.FUSE_NARROW_MIN_RATIO = 8, 0      # fuse every level the shapes allow, also the narrow last head
        G.FUSE_UP_PREDICT = True
        rec_f = n(model.decompress(coordinates=coords, strings=streams, shape=shapes, k=ks, q_vals=qs))
        G.FUSE_UP_PREDICT = False
        rec_u = n(model.decompress(coordinates=coords, strings=streams, shape=shapes, k=ks, q_vals=qs))
    finally:
        G.FUSE_UP_PREDICT, G.FUSE_MIN_HEAD_CHANNELS, G.FUSE_NARROW_MIN_RATIO = old_fuse, old_min, old_ratio
    assert rec_f.shape == rec_u.shape == (pc.shape[0], 6)
    assert np.array_equal(rec_f[:, :3], rec_u[:, :3])                             # same occupancy decisions
    lv_f, lv_u = np.rint(rec_f[:, 3:] * 255).astype(int), np.rint(rec_u[:, 3:] * 255).astype(int)
    assert np.abs(lv_f - lv_u).max() <= 1 and (lv_f != lv_u).mean() < 5e-3
    y_keys = n(coords[0]._pcc_cset.keys)[:coords[0].shape[0]]
    blk = dict(y_keys=y_keys, y_symbols=n(streams[0][0]), z_symbols=n(streams[0][1]), k=ks[0], q=q)
    rec_o = codec.decompress(P, cfg, [blk])
    assert np.array_equal(rec_f[:, :3], rec_o[:, :3])


def test_fused_path_edge_counts():
    """k = 0 at a level (nothing kept) and k larger than the candidate set (everything kept) through the composite
    up+head path: same result as the layer-wise path."""
    from unified_point_cloud_compression_amd import synth
    from unified_point_cloud_compression_amd.model.transforms import SparseSynthesisTransform as G
    cfg = codec.small_config()
    model = _model(cfg, codec.random_params(cfg, 2, gain=4.0))
    pc = synth.random_block(2, 24, 0.1)
    q = np.array([[0.5, 0.5]], dtype=np.float32)
    streams, shapes, ks, coords, qs = model.compress(t(pc), t(q), block_size=1024)
    old_fuse, old_min, old_ratio = G.FUSE_UP_PREDICT, G.FUSE_MIN_HEAD_CHANNELS, G.FUSE_NARROW_MIN_RATIO
    try:
        G.FUSE_MIN_HEAD_CHANNELS, G.FUSE_NARROW_MIN_RATIO = 8, 0
        for k_alt in ([[10 ** 9], [10 ** 9], ks[0][2]], [ks[0][0], [0], [0]]):
            out = {}
            for fuse in (True, False):
                G.FUSE_UP_PREDICT = fuse
                out[fuse] = n(model.decompress(coordinates=coords, strings=streams, shape=shapes, k=[k_alt], q_vals=qs))
            assert out[True].shape == out[False].shape
            assert np.array_equal(out[True][:, :3], out[False][:, :3])
    finally:
        G.FUSE_UP_PREDICT, G.FUSE_MIN_HEAD_CHANNELS, G.FUSE_NARROW_MIN_RATIO = old_fuse, old_min, old_ratio


def test_multi_block_partition_matches_oracle():
    from unified_point_cloud_compression_amd import synth
    cfg = codec.small_config()
    P = codec.random_params(cfg, 5, gain=4.0)
    model = _model(cfg, P)
    pc = synth.random_block(3, 48, 0.05)
    q = np.array([[0.5, 0.5]], dtype=np.float32)
    streams, shapes, ks, coords, qs = model.compress(t(pc), t(q), block_size=32)   # 2x2x2 blocks
    blocks = codec.compress(P, cfg, pc, q, block_size=32)
    assert len(streams) == len(blocks) == 8
    for i, b in enumerate(blocks):
        assert ks[i] == b["k"]
        assert np.array_equal(co.pack_keys(n(coords[i])), b["y_keys"])
    rec = n(model.decompress(coordinates=coords, strings=streams, shape=shapes, k=ks, q_vals=qs))
    assert rec.shape[0] == sum(b["n_points"] for b in blocks)


def test_deterministic_bitwise():
    """Encoder and decoder must see bit-identical h_s outputs (SURVEY section 7 'hard parts'): run twice, compare bits."""
    from unified_point_cloud_compression_amd import synth
    cfg = codec.small_config()
    model = _model(cfg, codec.random_params(cfg, 1, gain=4.0))
    pc = t(synth.random_block(1, 40, 0.08))
    q = t(np.array([[0.5, 0.5]], dtype=np.float32))
    a = model.compress(pc, q)
    b = model.compress(pc, q)
    assert torch.equal(a[0][0][0], b[0][0][0]) and torch.equal(a[0][0][1], b[0][0][1])
    ra = model.decompress(coordinates=a[3], strings=a[0], shape=a[1], k=a[2], q_vals=a[4])
    rb = model.decompress(coordinates=b[3], strings=b[0], shape=b[1], k=b[2], q_vals=b[4])
    assert torch.equal(ra, rb)
    # the range-guard fallback (g_a / g_s in the six-term form, hyper-synthesis pinned): the symbols of the hyper-prior path
    # do not move, the fallback is itself deterministic, and a stream coded under it decodes under the default form
    from unified_point_cloud_compression_amd import lib as L
    with L.arith_scope(L.ARITH_BF6):
        c = model.compress(pc, q)
        d = model.compress(pc, q)
        rc = model.decompress(coordinates=c[3], strings=c[0], shape=c[1], k=c[2], q_vals=c[4])
    assert torch.equal(c[0][0][0], d[0][0][0]) and torch.equal(c[0][0][1], d[0][0][1])
    rd = model.decompress(coordinates=c[3], strings=c[0], shape=c[1], k=c[2], q_vals=c[4])     # decoder in the default form
    with L.arith_scope(L.ARITH_BF6):
        assert torch.equal(rc, model.decompress(coordinates=d[3], strings=d[0], shape=d[1], k=d[2], q_vals=d[4]))
    assert rd.shape == rc.shape and torch.equal(rd[:, :3], rc[:, :3])                          # same geometry either way


def count_bits(strings):
    """`utils.count_bits` (`utils.py:30-48`)."""
    total = 0
    for st in strings:
        total += count_bits(st) if isinstance(st, list) else len(st) * 8
    return total


@pytest.mark.parametrize("coder", ["pcc_streams", "ans"])
@pytest.mark.parametrize("seed", [0, 2])
def test_real_bitstrings_roundtrip(seed, coder):
    """compress -> rANS byte strings -> decompress reproduces exactly what the symbol hand-off path reconstructs, and
    the string length matches the likelihood-based rate."""
    from unified_point_cloud_compression_amd import synth
    adaptive = seed == 2
    cfg = codec.small_config(adaptive=adaptive, offsets=adaptive)
    P = codec.random_params(cfg, seed, gain=make_golden.GAIN[seed])
    pc, q = t(synth.random_block(seed, 32, 0.08)), t(np.array([[0.4, 0.6]], dtype=np.float32))
    ref_model = _model(cfg, P, "symbols")
    a = ref_model.compress(pc, q)
    rec_ref = ref_model.decompress(coordinates=a[3], strings=a[0], shape=a[1], k=a[2], q_vals=a[4])
    model = _model(cfg, P, coder)
    b = model.compress(pc, q)
    (y_string,), (z_string,) = b[0][0]
    assert isinstance(y_string, bytes) and isinstance(z_string, bytes)
    rec = model.decompress(coordinates=b[3], strings=b[0], shape=b[1], k=b[2], q_vals=b[4])
    assert torch.equal(rec, rec_ref)
    # rate: coded bits vs -sum log2(likelihood)
    x = model.block_input(pc)
    y, _ = model.g_a(x)
    y_lik, z_lik = model.entropy_model.likelihoods(y, q)
    ideal = float(-(torch.log2(y_lik.double()).sum() + torch.log2(z_lik.double()).sum()))
    bits = count_bits(b[0])
    overhead = 8 * 4 * (2 + y_lik.shape[1] + z_lik.shape[1]) + 2 * 64 * (y_lik.shape[1] + z_lik.shape[1]) if coder == "pcc_streams" else 256
    # 16-bit tables floor every probability at 2^-16 (+ bypass digits), the likelihood floor is 1e-9: with random
    # weights (poor model fit) the coder can be cheaper than the ideal, never much more expensive
    assert ideal * 0.5 < bits < ideal * 1.05 + overhead, (bits, ideal)


def test_gpu_streams_equal_host_coder_per_channel():
    """Each per-channel GPU stream is byte-identical to the host single-stream coder run on that channel alone, and
    the oracle's pure-Python coder agrees with both."""
    from oracle import rans
    from unified_point_cloud_compression_amd.compressai.entropy_models import GaussianConditional, get_scale_table
    rng = np.random.default_rng(0)
    rows, c = 700, 6
    gc = GaussianConditional(None).to(dev())
    gc.update_scale_table(get_scale_table(), force=True)
    idx = rng.integers(0, 64, (rows, c)).astype(np.int32)
    st = n(gc.scale_table)
    sym = np.rint(rng.standard_normal((rows, c)) * st[idx] * 1.3).astype(np.int32)
    sym[::41, 0] += 9000
    sym[5::77, 3] = -123456
    gc.STREAM_SYMBOLS = 1                                           # one stream per channel
    data = gc.compress_rows(t(sym), t(idx))
    w = np.frombuffer(data, "<u4")
    assert w[0] == c
    lens = w[1:1 + c]
    assert len(w) == 1 + c + lens.sum()
    cdf, sizes, offs = (n(x) for x in (gc._quantized_cdf, gc._cdf_length, gc._offset))
    off = 1 + c
    host = GaussianConditional(None, entropy_coder="ans")
    host._quantized_cdf, host._cdf_length, host._offset = gc._quantized_cdf.cpu(), gc._cdf_length.cpu(), gc._offset.cpu()
    for ch in range(c):
        stream = w[off:off + lens[ch]].tobytes()
        off += lens[ch]
        assert stream == host._host_encode(sym[:, ch].copy(), idx[:, ch].copy())
        assert stream == rans.encode(sym[:, ch], idx[:, ch], cdf, sizes, offs)
    back = gc.decompress_rows(data, rows, c, t(idx))
    assert np.array_equal(n(back), sym)
    # grouped streams (several channels per stream, row by row) round-trip too and carry less framing
    rows2, c2 = 300, 8
    idx2 = rng.integers(0, 64, (rows2, c2)).astype(np.int32)
    sym2 = np.rint(rng.standard_normal((rows2, c2)) * st[idx2]).astype(np.int32)
    gc.STREAM_SYMBOLS = 1200
    assert gc.n_streams(rows2, c2) == (2, 1)
    d2 = gc.compress_rows(t(sym2), t(idx2))
    assert np.frombuffer(d2, "<u4")[0] == 2
    assert np.array_equal(n(gc.decompress_rows(d2, rows2, c2, t(idx2))), sym2)
    host4 = np.frombuffer(d2, "<u4")
    grp = sym2[:, :4].reshape(-1), idx2[:, :4].reshape(-1)           # stream 0 = channels 0..3, row by row
    assert host4[3:3 + host4[1]].tobytes() == host._host_encode(grp[0].copy(), grp[1].copy())
    gc.STREAM_SYMBOLS = 1
    # row segments: stream (segment, channel) = host coder on that tile; ragged last segment
    for seg_symbols, want in ((200, 3), (64, 10)):
        gc.SEGMENT_SYMBOLS = seg_symbols
        ng, segs = gc.n_streams(rows, c)
        assert (ng, segs) == (c, want)
        d3 = gc.compress_rows(t(sym), t(idx))
        w3 = np.frombuffer(d3, "<u4")
        assert w3[0] == ng * segs and len(w3) == 1 + ng * segs + w3[1:1 + ng * segs].sum()
        R = -(-rows // segs)
        o3 = 1 + ng * segs
        for sgi in range(segs):
            for ch in range(c):
                ln = w3[1 + sgi * ng + ch]
                tile = slice(sgi * R, min(rows, (sgi + 1) * R))
                assert w3[o3:o3 + ln].tobytes() == host._host_encode(sym[tile, ch].copy(), idx[tile, ch].copy())
                o3 += ln
        assert np.array_equal(n(gc.decompress_rows(d3, rows, c, t(idx))), sym)
    # 9 rows in 4 segments of 3: the last segment is empty (two state words per stream)
    gc.SEGMENT_SYMBOLS = 2
    assert gc.n_streams(9, c) == (c, 4)
    d4 = gc.compress_rows(t(sym[:9]), t(idx[:9]))
    assert np.array_equal(n(gc.decompress_rows(d4, 9, c, t(idx[:9]))), sym[:9])
    gc.SEGMENT_SYMBOLS = 1 << 30
    # a truncated / corrupted container is reported, not decoded
    from unified_point_cloud_compression_amd import lib as L
    bad = bytearray(data)
    bad[0] ^= 0xFF
    with pytest.raises(L.PccError):
        gc.decompress_rows(bytes(bad), rows, c, t(idx))


def test_mutated_stream_containers_are_rejected_or_decoded_never_read_out_of_bounds():
    """VERDICT r3 item 7, device side: 300 seeded mutations (truncation to whole words, bit flips, lying stream lengths /
    stream count, noise) of a `pcc_streams` container through the GPU decoder.  Its reads are clamped to the uploaded buffer
    and every stream's extent is checked against the container, so the outcome is a status word (-> PccError) or symbols of
    the right shape; the single-stream host coder and the octree coder are fuzzed under ASan in tests/test_cpu_fuzz.py."""
    from unified_point_cloud_compression_amd import lib as L
    from unified_point_cloud_compression_amd.compressai.entropy_models import GaussianConditional, get_scale_table
    rng = np.random.default_rng(11)
    gc = GaussianConditional(None).to(dev())
    gc.update_scale_table(get_scale_table(), force=True)
    rows, c = 500, 16
    idx = rng.integers(0, 64, (rows, c)).astype(np.int32)
    sym = np.rint(rng.standard_normal((rows, c)) * n(gc.scale_table)[idx]).astype(np.int32)
    data = gc.compress_rows(t(sym), t(idx))
    assert np.array_equal(n(gc.decompress_rows(data, rows, c, t(idx))), sym)
    words = np.frombuffer(data, "<u4").copy()
    ok = bad = 0
    for case in range(300):
        w = words.copy()
        kind = case % 5
        if kind == 0:
            w = w[:int(rng.integers(1, len(w)))]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 9))):
                w[int(rng.integers(0, len(w)))] ^= np.uint32(1 << int(rng.integers(0, 32)))
        elif kind == 2:
            w[int(rng.integers(1, 1 + int(words[0])))] = np.uint32(rng.choice([0, 1, 2, 0xFFFFFFFF, 0x7FFFFFFF, int(rng.integers(0, 1 << 20))]))
        elif kind == 3:
            w[0] = np.uint32(rng.choice([0, 1, int(words[0]) * 2, 0xFFFFFFFF]))
        else:
            w = rng.integers(0, 1 << 32, int(rng.integers(1, 200)), dtype=np.uint64).astype(np.uint32)
        try:
            out = gc.decompress_rows(w.tobytes(), rows, c, t(idx))
            assert tuple(out.shape) == (rows, c)
            ok += 1
        except L.PccError:
            bad += 1
    torch.cuda.synchronize()
    assert ok + bad == 300 and bad >= 100, (ok, bad)


def test_file_bitstream_roundtrip_and_rd_figures(tmp_path):
    """compress(path=...) -> file -> decompress(path=...) equals the in-memory path; file bpp (`utils.py:471`) and
    D1-PSNR (`metrics/metric.py:113-119`) are identical for the HIP path and the oracle (same geometry)."""
    import os
    from unified_point_cloud_compression_amd import synth, metrics
    cfg = codec.small_config()
    P = codec.random_params(cfg, 0, gain=4.0)
    model = _model(cfg, P, "pcc_streams")
    pc_np = synth.random_block(4, 48, 0.06)
    pc, q = t(pc_np), t(np.array([[0.5, 0.5]], dtype=np.float32))
    mem = model.compress(pc, q, block_size=32)                     # 8 blocks
    rec_mem = model.decompress(coordinates=mem[3], strings=mem[0], shape=mem[1], k=mem[2], q_vals=mem[4])
    path = os.path.join(tmp_path, "bitstream.bin")
    assert model.compress(pc, q, path=path, block_size=32) is None
    rec_file = model.decompress(path=path)
    assert torch.equal(rec_file, rec_mem)
    n_pts = pc_np.shape[0]
    bpp_file = os.path.getsize(path) * 8 / n_pts
    bpp_strings = metrics.count_bits(mem[0]) / n_pts
    assert bpp_file > bpp_strings                                   # + coordinates + 44-byte block headers
    coord_bits = sum(len(model.gpcc_encode(c)) for c in mem[3]) * 8
    # format word (8 bytes) + block count (4) + per block: the reference's 44-byte header + the 4-byte stream-geometry word
    assert abs(os.path.getsize(path) * 8 - (metrics.count_bits(mem[0]) + coord_bits + 8 * (8 + 4 + 48 * len(mem[0])))) == 0
    # D1 parity with the oracle: decode the same symbols with the oracle and compare the PSNR figure
    sym_model = _model(cfg, P, "symbols")
    a = sym_model.compress(pc, q, block_size=32)
    blocks = [dict(y_keys=n(c._pcc_cset.keys)[:c.shape[0]], y_symbols=n(s[0]), z_symbols=n(s[1]), k=k, q=n(q))
              for c, s, k in zip(a[3], a[0], a[2])]
    rec_o = codec.decompress(P, cfg, blocks)
    src = np.floor(pc_np[:, :3])
    from oracle import metrics as ometrics
    d_gpu = metrics.d1_psnr(t(src.astype(np.float32)), rec_file[:, :3], resolution=63)
    d_ora = ometrics.d1_psnr(src, rec_o[:, :3], resolution=63)
    assert all(abs(x - y) <= 1e-9 * abs(y) for x, y in zip(d_gpu, d_ora))
    # the full report (geometry + colour, both directions) against the oracle
    rep_g = metrics.pointcloud_metrics(pc, rec_file, resolution=63)
    rep_o = ometrics.pointcloud_metrics(pc_np, n(rec_file), resolution=63)
    assert set(rep_g) == set(rep_o)
    for k in rep_o:
        assert abs(rep_g[k] - rep_o[k]) <= 1e-6 * max(1.0, abs(rep_o[k])), k


def test_nearest_neighbour_search_exact():
    """pcc_nn_sorted_x against the oracle: random clouds, far outliers, duplicates of the query set, ties (smallest
    canonical row wins), a one-point target."""
    from oracle import metrics as ometrics
    from unified_point_cloud_compression_amd import metrics
    rng = np.random.default_rng(5)
    for na, nb, span in ((5000, 4000, 64), (3000, 7000, 300), (100, 1, 50), (2000, 2000, 8)):
        a = np.unique(rng.integers(0, span, (na, 3)), axis=0)
        b = np.unique(rng.integers(0, span, (nb, 3)), axis=0)
        if nb > 1:
            b = np.unique(np.concatenate([b, [[span * 4, 0, 0], [0, span * 4, span * 4]]]), axis=0)   # outliers
        d2o, nno = ometrics.nearest(a, b)
        d2, nn = metrics.nearest(t(a.astype(np.int32)), t(b.astype(np.int32)))
        assert np.array_equal(n(d2), d2o)
        assert np.array_equal(n(nn), nno)


def test_stream_count_adapts_to_the_payload_and_stays_reproducible():
    """The encoder raises the stream count of a large symbol matrix until the 12-byte framing per stream reaches 2 % of
    the payload it estimates from the symbols (`pcc_rans_estimate_bits`); the decoder reads the count in the container.
    Same symbols -> same bytes; a low-rate matrix keeps the (n, c) rule's count."""
    from unified_point_cloud_compression_amd.compressai.entropy_models import GaussianConditional, get_scale_table
    rng = np.random.default_rng(4)
    rows, c = 6000, 64                                             # 384 k symbols: above ADAPTIVE_MIN_SYMBOLS
    gc = GaussianConditional(None).to(dev())
    gc.update_scale_table(get_scale_table(), force=True)
    st = n(gc.scale_table)
    idx_hi = rng.integers(30, 50, (rows, c)).astype(np.int32)      # wide scales: many bits per symbol
    sym_hi = np.rint(rng.standard_normal((rows, c)) * st[idx_hi]).astype(np.int32)
    sym_hi[::97, 3] = 40000                                         # escapes (bypass digits) in the estimate too
    idx_lo = np.zeros((rows, c), np.int32)                          # narrowest scale, mostly zeros: ~0.1 bit per symbol
    sym_lo = (rng.random((rows, c)) < 0.01).astype(np.int32)
    base_ng, base_segs = gc.n_streams(rows, c)
    d_hi = gc.compress_rows(t(sym_hi), t(idx_hi))
    d_lo = gc.compress_rows(t(sym_lo), t(idx_lo))
    ns_hi, ns_lo = int(np.frombuffer(d_hi[:4], "<u4")[0]), int(np.frombuffer(d_lo[:4], "<u4")[0])
    assert ns_lo == base_ng * base_segs                             # low rate: the (n, c) rule stands
    assert ns_hi > base_ng * base_segs and ns_hi % base_ng == 0     # high rate: more segments
    framing = 12 * ns_hi / len(d_hi)
    assert framing <= 0.03, framing                                 # target 2 % of the payload (estimate within a few %)
    assert rows * c // ns_hi >= gc.MIN_SEGMENT_SYMBOLS
    assert np.array_equal(n(gc.decompress_rows(d_hi, rows, c, t(idx_hi))), sym_hi)
    assert np.array_equal(n(gc.decompress_rows(d_lo, rows, c, t(idx_lo))), sym_lo)
    assert gc.compress_rows(t(sym_hi), t(idx_hi)) == d_hi           # reproducible bytes
    est = abs(len(d_hi) - 12 * ns_hi)                               # the estimate the count was derived from was close:
    assert 0.6 <= (ns_hi // base_ng) * base_ng * 12 / (0.02 * est) <= 1.05 or rows * c // ns_hi <= 2 * gc.MIN_SEGMENT_SYMBOLS
