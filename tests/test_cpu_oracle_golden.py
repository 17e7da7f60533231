"""CPU suite: the oracle reproduces the committed goldens; entropy formulas in closed form; host logic."""
import numpy as np
import pytest
from scipy.stats import norm

from oracle import codec, entropy as en, coords as co
from tests.golden import make_golden


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_reproduces_golden(seed):
    d = make_golden.case(seed)
    g = np.load(f"tests/golden/codec_seed{seed}.npz")
    for key in ("k", "y_keys", "z_keys", "y_symbols", "z_symbols", "indexes", "mask_0", "mask_1", "mask_2", "keys_2"):
        assert np.array_equal(d[key], g[key]), key
    assert np.array_equal(d["recon"][:, :3], g["recon"][:, :3])
    assert np.abs(d["recon"][:, 3:] - g["recon"][:, 3:]).max() <= 1e-6
    assert abs(float(d["bits"]) - float(g["bits"])) < 1e-6 * float(g["bits"])


def test_gaussian_likelihood_closed_form():
    v = np.array([0, 1, -3, 7], dtype=np.float32)
    s = np.array([0.05, 0.5, 2.0, 300.0], dtype=np.float32)
    got = en.gaussian_likelihood(v, s)
    sb = np.maximum(s, 0.11).astype(np.float64)
    want = norm.cdf((0.5 - np.abs(v)) / sb) - norm.cdf((-0.5 - np.abs(v)) / sb)
    np.testing.assert_allclose(got, np.maximum(want, 1e-9), rtol=2e-5, atol=1e-8)


def test_scale_table_and_indexes():
    tab = en.scale_table()
    assert len(tab) == 64 and abs(tab[0] - 0.11) < 1e-6 and abs(tab[-1] - 256) < 1e-3
    s = np.array([0.0, 0.11, 0.1100001, tab[10], tab[10] * 1.0001, 255.9, 256.0, 1e4], dtype=np.float32)
    idx = en.build_indexes(s)
    assert idx.tolist() == [0, 0, 1, 10, 11, 63, 63, 63]


def test_entropy_bottleneck_is_a_distribution():
    p = en.eb_init(3, seed=0)
    v = np.arange(-400, 401, dtype=np.float32)[None, :].repeat(3, 0)
    lik = en.eb_likelihood(p, v)
    assert np.all(lik >= 1e-9) and np.allclose(lik.sum(axis=1), 1.0, atol=1e-3)
    sym, zh = en.eb_quantize(p, np.array([[0.4, 0.6, -1.5]] * 3, dtype=np.float32))
    assert sym[0].tolist() == [0, 1, -2]          # round half to even, median 0


def test_partition_blocks_order_and_counts():
    rng = np.random.default_rng(0)
    pc = np.concatenate([rng.integers(0, 64, (500, 3)).astype(np.float32), rng.random((500, 3), dtype=np.float32)], 1)
    order, counts = codec.partition_blocks(pc, 32)
    assert counts.sum() == 500 and len(counts) == 8
    bi = np.floor((pc[order, :3] - pc[:, :3].min(0)) / 32).astype(int)
    code = bi[:, 0] * 4 + bi[:, 1] * 2 + bi[:, 2]
    assert np.all(np.diff(code) >= 0)
