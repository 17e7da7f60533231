"""Rate / distortion figures the reference's harness reports (evaluation utility, host side, not on the hot path):
D1 (point-to-point) PSNR as in `metrics/metric.py:113-119,74` and bits per input point as in `utils.py:471`."""
import numpy as np
from scipy.spatial import cKDTree


def d1_psnr(a_xyz, b_xyz, resolution=1023):
    """10*log10(res^2 / mean_i(||a_i - nn_B(a_i)||^2 / 3)); returns (A->B, B->A, symmetric = min)."""
    a, b = np.asarray(a_xyz, dtype=np.float64), np.asarray(b_xyz, dtype=np.float64)

    def one(p, q):
        d, _ = cKDTree(q).query(p, k=1)
        mse = float(np.mean(d ** 2 / 3.0))
        return float("inf") if mse == 0 else 10.0 * np.log10(resolution ** 2 / mse)
    ab, ba = one(a, b), one(b, a)
    return ab, ba, min(ab, ba)


def count_bits(strings):
    """`utils.count_bits` (`utils.py:30-48`)."""
    return sum(count_bits(s) if isinstance(s, list) else len(s) * 8 for s in strings)
