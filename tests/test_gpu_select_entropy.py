"""a4 / a6 / a7 / a8: top-k masks & pruning (bit exact), lookup gather, entropy-model kernels."""
import numpy as np
import pytest
import torch

from oracle import coords as co, ops, entropy as en
from tests.util import dev, t, n, cloud_keys, assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,k", [(10, 3), (1000, 1), (5000, 4999), (100_000, 20_000), (300_000, 150_000), (50, 50), (50, 0)])
def test_topk_mask_bit_exact(rows, k):
    from unified_point_cloud_compression_amd import sparse as S
    rng = np.random.default_rng(rows + k)
    logits = rng.standard_normal(rows).astype(np.float32)
    logits[rng.integers(0, rows, rows // 3)] = np.float32(0.25)      # many exact ties at one value
    logits[rng.integers(0, rows, max(rows // 50, 1))] = np.float32(-0.0)
    got = S.topk_mask(t(logits)[:, None], [0, rows], [k])
    want = ops.topk_mask(logits, [k])
    assert np.array_equal(n(got), want)
    assert int(n(got).sum()) == min(k, rows)


def test_topk_threshold_inside_tie_run_and_batches():
    from unified_point_cloud_compression_amd import sparse as S
    logits = np.array([1, 5, 5, 5, 5, 0, 5, 9, -3, 5], dtype=np.float32)
    for k in range(0, 11):
        got = n(S.topk_mask(t(logits)[:, None], [0, 10], [k]))
        assert np.array_equal(got, ops.topk_mask(logits, [k])), k
    # two batches, independent k
    batch = np.array([0] * 6 + [1] * 4)
    got = n(S.topk_mask(t(logits)[:, None], [0, 6, 10], [2, 3]))
    assert np.array_equal(got, ops.topk_mask(logits, [2, 3], batch))


@pytest.mark.parametrize("rows", [10, 2047, 2048, 2049, 100_000, 1_300_000])
def test_topk_prune_keys_equals_mask_then_prune(rows):
    """`pcc_topk_prune_keys` (selection + key compaction in one pass: the decoder's composite levels) against the oracle's
    mask and `keys[mask]`: ties at the threshold inside and across the 2048-row workgroups, strided logits, three batches
    with k = 0 / k inside a tie run / k >= rows."""
    from unified_point_cloud_compression_amd import sparse as S
    rng = np.random.default_rng(rows)
    logits = rng.standard_normal(rows).astype(np.float32)
    logits[rng.integers(0, rows, rows // 2)] = np.float32(0.125)      # half the rows tie at one value
    logits[rng.integers(0, rows, max(rows // 40, 1))] = np.float32(-0.0)
    keys = np.sort(rng.choice(10 ** 12, rows, replace=False)).astype(np.int64)
    wide = np.stack([logits, rng.standard_normal(rows).astype(np.float32)], axis=1)       # column 0 of an [n, 2] tensor
    for k in sorted({0, 1, rows // 3, rows // 2, rows - 1, rows, rows + 5}):
        want = ops.topk_mask(logits, [k])
        mask, ko, cnt = S.topk_prune_keys(t(wide), [0, rows], [k], t(keys))
        assert cnt == min(k, rows) == int(want.sum()), k
        assert np.array_equal(n(mask), want), k
        assert np.array_equal(n(ko), keys[want]), k
        assert np.array_equal(n(S.topk_mask(t(logits)[:, None], [0, rows], [k])), want), k
    if rows >= 100:
        cut = [0, rows // 5, rows // 2, rows]
        batch = np.repeat(np.arange(3), np.diff(cut))
        ks = [0, (cut[2] - cut[1]) // 2, rows]
        want = ops.topk_mask(logits, ks, batch)
        mask, ko, cnt = S.topk_prune_keys(t(logits)[:, None], cut, ks, t(keys))
        assert np.array_equal(n(mask), want) and np.array_equal(n(ko), keys[want]) and cnt == int(want.sum())


def test_topk_many_segments_in_one_call():
    """Round 4: all batch segments of a call share the launches (segment = grid y, 48 per batch of launches).  130 segments of
    ragged sizes -- empty ones, k = 0, k >= rows, ties -- against the oracle, mask and compacted keys, so that the chunking over
    more than 48 segments and the per-segment output offsets are covered."""
    from unified_point_cloud_compression_amd import sparse as S
    rng = np.random.default_rng(130)
    sizes = rng.integers(0, 5000, 130)
    sizes[[3, 50, 129]] = 0
    cut = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    rows = int(cut[-1])
    logits = rng.standard_normal(rows).astype(np.float32)
    logits[rng.integers(0, rows, rows // 4)] = np.float32(0.5)
    ks = [int(rng.integers(0, s + 3)) if s else int(rng.integers(0, 3)) for s in sizes]
    ks[7], ks[60] = 0, int(sizes[60]) + 9
    batch = np.repeat(np.arange(130), sizes)
    keys = np.sort(rng.choice(10 ** 12, rows, replace=False)).astype(np.int64)
    want = ops.topk_mask(logits, ks, batch)
    mask, ko, cnt = S.topk_prune_keys(t(logits)[:, None], cut.tolist(), ks, t(keys))
    assert cnt == int(want.sum())
    assert np.array_equal(n(mask), want)
    assert np.array_equal(n(ko), keys[want])
    assert np.array_equal(n(S.topk_mask(t(logits)[:, None], cut.tolist(), ks)), want)


def test_prune_rows():
    from unified_point_cloud_compression_amd import sparse as S
    rng = np.random.default_rng(1)
    for rows, c in ((5000, 128), (777, 3), (1, 4), (4096, 32)):
        keys = np.sort(rng.choice(10 ** 9, rows, replace=False)).astype(np.int64)
        f = rng.standard_normal((rows, c)).astype(np.float32)
        mask = rng.random(rows) < 0.3
        ko, fo, cnt = S.prune(t(keys), rows, t(f), t(mask))
        assert cnt == mask.sum()
        assert np.array_equal(n(ko), keys[mask]) and np.array_equal(n(fo), f[mask])
    ko, fo, cnt = S.prune(t(keys), rows, t(f), t(np.zeros(rows, dtype=bool)))     # all-False corner case (A.6)
    assert cnt == 0 and ko.shape[0] == 0


def test_lookup_gather_and_features_at_coordinates():
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    keys = cloud_keys(5, 20, 0.2, 8)
    rng = np.random.default_rng(2)
    f = rng.standard_normal((len(keys), 256)).astype(np.float32)
    x = ME.SparseTensor(coordinates=t(co.unpack_keys(keys)), features=t(f), tensor_stride=8)
    qk = np.concatenate([keys[::3], keys[:5] + 1, keys[-3:] + (1 << 40)])       # present, off-grid-by-1, absent
    q = co.unpack_keys(qk).astype(np.float32)
    got = x.features_at_coordinates(t(q))
    assert_close(n(got), ops.features_at(keys, f, qk), atol=0, rtol=0, what="features_at_coordinates")


@pytest.mark.parametrize("adaptive", [False, True])
def test_gaussian_conditional_kernels(adaptive):
    from unified_point_cloud_compression_amd.compressai.entropy_models import GaussianConditional, get_scale_table
    rng = np.random.default_rng(3)
    rows, c, nb = 3000, 128, 2
    y = (rng.standard_normal((rows, c)) * 6).astype(np.float32)
    scales = np.exp(rng.uniform(np.log(0.01), np.log(300), (rows, c))).astype(np.float32)
    tab = en.scale_table()
    scales[:64, 0] = tab                                  # exactly on table entries (<= boundary)
    means = rng.standard_normal((rows, c)).astype(np.float32)
    params = np.concatenate([scales, means], axis=1)
    keys = np.sort(rng.choice(10 ** 6, rows, replace=False)).astype(np.int64)
    keys[rows // 2:] += (1 << 48)
    gain = (0.5 + rng.random((nb, c))).astype(np.float32) if adaptive else None
    g_rows = gain[(keys >> 48)] if adaptive else np.ones((rows, c), np.float32)
    gc = GaussianConditional(None).to(dev())
    gc.update_scale_table(get_scale_table(), force=True)
    assert np.allclose(n(gc.scale_table), tab, rtol=1e-6)
    sym, idx, lik = gc.encode_rows(t(y), t(params), t(keys), t(gain) if adaptive else None)
    want_sym = en.quantize_symbols(y * g_rows, means * g_rows)
    want_idx = en.build_indexes(scales * g_rows, n(gc.scale_table))
    assert np.array_equal(n(sym), want_sym)
    assert np.array_equal(n(idx), want_idx)
    want_lik = en.gaussian_likelihood(want_sym.astype(np.float32), scales * g_rows)
    assert_close(n(lik), want_lik, atol=1e-6, rtol=1e-4, what="gaussian likelihood")
    y_hat, idx2 = gc.decode_rows(sym, t(params), t(keys), t(gain) if adaptive else None)
    assert np.array_equal(n(idx2), want_idx)                                   # encoder/decoder index identity
    assert_close(n(y_hat), en.dequantize(want_sym, means * g_rows), atol=1e-6, rtol=1e-6, what="y_hat")
    # [B,C,N] surface used by the reference (`model/entropy_models.py:396`)
    idx3 = gc.build_indexes(t(scales * g_rows).t().unsqueeze(0))
    assert np.array_equal(n(idx3)[0].T, want_idx)


def test_entropy_bottleneck_kernel():
    from unified_point_cloud_compression_amd.compressai.entropy_models import EntropyBottleneck
    rng = np.random.default_rng(4)
    c, rows = 192, 900
    eb = EntropyBottleneck(c).to(dev())
    p = en.eb_init(c, seed=7)
    for i in range(5):
        p[f"_matrix{i}"] += rng.normal(0, 0.3, p[f"_matrix{i}"].shape).astype(np.float32)
        if i < 4:
            p[f"_factor{i}"] += rng.normal(0, 0.3, p[f"_factor{i}"].shape).astype(np.float32)
    p["quantiles"][:, 0, 1] = rng.normal(0, 2, c).astype(np.float32)
    with torch.no_grad():
        for k, v in p.items():
            getattr(eb, k).copy_(t(v))
    z = (rng.standard_normal((rows, c)) * 8).astype(np.float32)
    sym, zh, lik = eb.encode_rows(t(z))
    want_sym, want_zh = en.eb_quantize(p, z.T)
    assert np.array_equal(n(sym), want_sym.T)
    assert_close(n(zh), want_zh.T, atol=1e-6, rtol=1e-6, what="z_hat")
    assert_close(n(lik), en.eb_likelihood(p, want_zh).T, atol=1e-6, rtol=2e-4, what="z likelihood")
    # torch restatement of CompressAI's _logits_cumulative inside the shim agrees with the kernel's packing
    with torch.no_grad():
        v = zh.t().unsqueeze(1)
        lo, up = eb._logits_cumulative(v - 0.5), eb._logits_cumulative(v + 0.5)
        sg = -torch.sign(lo + up)
        ref = torch.abs(torch.sigmoid(sg * up) - torch.sigmoid(sg * lo))[:, 0, :].t().clamp_min(1e-9)
    assert_close(n(lik), n(ref), atol=1e-6, rtol=2e-4, what="z likelihood vs torch")
