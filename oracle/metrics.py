"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU restatement of the reference's distortion report:
`PointCloudMetric(...).compute_pointcloud_metrics(drop_duplicates=True)` as called at `evaluate.py:165-166`
(`metrics/metric.py:6-188`), with scipy's KD-tree in place of Open3D's.  Ties between equidistant neighbours are
resolved to the smallest row in (x,y,z) order, as the HIP path does (the reference leaves it to the KD-tree).
PARITY UNPINNED: the reference ships no fixtures for this report."""
import numpy as np
from scipy.spatial import cKDTree


def _dedup_sorted(xyz, rgb=None):
    """`remove_duplicated_points` (`metric.py:19-21`; first occurrence wins), then rows in (x,y,z) order."""
    xyz = np.floor(np.asarray(xyz, dtype=np.float64)).astype(np.int64)
    _, first = np.unique(xyz, axis=0, return_index=True)          # np.unique sorts rows lexicographically
    return xyz[first], (None if rgb is None else np.asarray(rgb, dtype=np.float64)[first])


def nearest(a, b):
    """(squared distance, row of b) of the nearest b to each a; smallest row among equidistant ones."""
    tree = cKDTree(b.astype(np.float64))
    k = min(16, len(b))
    d, j = tree.query(a.astype(np.float64), k=k)
    d, j = d.reshape(len(a), -1), j.reshape(len(a), -1)
    d2 = np.rint(d ** 2).astype(np.int64)
    best = d2[:, 0]
    tie = d2 == best[:, None]
    if k > 1 and np.any(tie[:, -1]) and k < len(b):               # more ties than neighbours fetched: exact fallback
        for i in np.nonzero(tie[:, -1])[0]:
            e = ((b - a[i]) ** 2).sum(1)
            j[i, 0] = int(np.flatnonzero(e == e.min())[0])
            tie[i, 1:] = False
    cand = np.where(tie, j, np.iinfo(np.int64).max)
    return best, cand.min(axis=1)


def rgb_to_yuv(rgb):
    """`convert_rgb_to_yuv` (`metric.py:170-188`), inputs in [0,1]."""
    c = (np.asarray(rgb) * 255).astype(np.uint8)
    yuv = np.empty(c.shape, dtype=np.float32)
    yuv[..., 0] = 0.2126 * c[..., 0] + 0.7152 * c[..., 1] + 0.0722 * c[..., 2]
    yuv[..., 1] = -0.1146 * c[..., 0] - 0.3854 * c[..., 1] + 0.5 * c[..., 2]
    yuv[..., 2] = 0.5 * c[..., 0] - 0.4542 * c[..., 1] - 0.0458 * c[..., 2]
    yuv = yuv / 255.0
    yuv[..., 1] += 0.5
    yuv[..., 2] += 0.5
    return yuv


def _psnr(peak2, mse):
    return float("inf") if mse == 0 else float(10 * np.log10(peak2 / mse))


def _one_direction(prefix, a, a_rgb, b, b_rgb, resolution):
    d2, nn = nearest(a, b)
    l2 = d2 / 3.0
    r = {prefix + "mse": float(l2.mean()), prefix + "hausdorff": float(l2.max())}
    r[prefix + "psnr_mse"] = _psnr(resolution ** 2, r[prefix + "mse"])
    r[prefix + "psnr_hausdorff"] = _psnr(resolution ** 2, r[prefix + "hausdorff"])
    if a_rgb is not None and b_rgb is not None:
        rnd = lambda c: np.clip(np.round(c * 255.0) / 255.0, 0.0, 1.0)      # noqa: E731
        e = ((rgb_to_yuv(rnd(a_rgb)) - rgb_to_yuv(rnd(b_rgb[nn]))) ** 2).astype(np.float64).mean(axis=0)
        for i, ch in enumerate("yuv"):
            r[prefix + ch + "_mse"] = float(e[i])
            r[prefix + ch + "_psnr"] = _psnr(1.0, float(e[i]))
        r[prefix + "yuv_mse"] = float(e.mean())
        r[prefix + "yuv_psnr"] = _psnr(1.0, float(e.mean()))
    return r


def pointcloud_metrics(source, reconstruction, resolution=1023):
    sa, sb = np.asarray(source), np.asarray(reconstruction)
    a, a_rgb = _dedup_sorted(sa[:, :3], sa[:, 3:6] if sa.shape[1] >= 6 else None)
    b, b_rgb = _dedup_sorted(sb[:, :3], sb[:, 3:6] if sb.shape[1] >= 6 else None)
    r = {}
    r.update(_one_direction("AB_", a, a_rgb, b, b_rgb, resolution))
    r.update(_one_direction("BA_", b, b_rgb, a, a_rgb, resolution))
    keys = ["mse", "hausdorff", "psnr_mse", "psnr_hausdorff"]                 # `metric.py:72-83`
    if a_rgb is not None and b_rgb is not None:
        keys += [c + s for c in "yuv" for s in ("_mse", "_psnr")]
    for k in keys:
        r["sym_" + k] = min(r["AB_" + k], r["BA_" + k])
    return r


def d1_psnr(a_xyz, b_xyz, resolution=1023):
    """10*log10(res^2 / mean_i(||a_i - nn_B(a_i)||^2 / 3)); returns (A->B, B->A, symmetric = min)."""
    r = pointcloud_metrics(np.asarray(a_xyz)[:, :3], np.asarray(b_xyz)[:, :3], resolution)
    return r["AB_psnr_mse"], r["BA_psnr_mse"], r["sym_psnr_mse"]
