"""Rate / distortion figures the reference's harness reports (`evaluate.py:165-166`): the `PointCloudMetric` report
(`metrics/metric.py:6-188`) and bits per input point (`utils.py:471`, `utils.py:30-48`).

The nearest-neighbour association -- two Open3D KD-trees and a Python loop per point in the reference -- runs on the
GPU (`pcc_nn_sorted_x`); the per-point arithmetic after it is a handful of float64 tensor ops on the device."""
import math

import torch

from . import lib as L
from . import sparse as S

_YUV = ((0.2126, 0.7152, 0.0722), (-0.1146, -0.3854, 0.5), (0.5, -0.4542, -0.0458))     # BT.709, `metric.py:179-181`


def _canonical(xyz, rgb=None):
    """Voxel cloud -> (int32 xyz sorted by (x,y,z), rgb in the same order), duplicates dropped (first wins), like
    `remove_duplicated_points` at `metrics/metric.py:19-21`."""
    xyz = torch.as_tensor(xyz)
    if not xyz.is_cuda:
        raise L.PccError("metrics: GPU tensors required (no CPU fallback)")
    n = xyz.shape[0]
    coords = torch.cat([torch.zeros((n, 1), device=xyz.device, dtype=xyz.dtype), xyz[:, :3]], dim=1)
    cs, perm, keep = S.coordset_from_coords(coords, 1)
    pts = cs.coords()[:, 1:].contiguous()
    if rgb is None:
        return pts, None
    rgb = torch.as_tensor(rgb, device=xyz.device)
    if keep is not None:
        rgb = rgb[keep]
    if perm is not None:        # perm: canonical position -> row of the de-duplicated user-order tensor
        rgb = rgb[perm]
    return pts, rgb.to(torch.float64)


def nearest(a_xyz, b_xyz_sorted):
    """For every row of a (int32 [n,3]) the squared distance to, and the row of, its nearest point of b (sorted by x)."""
    a = a_xyz.to(torch.int32).contiguous()
    b = b_xyz_sorted.to(torch.int32).contiguous()
    d2 = torch.empty(a.shape[0], dtype=torch.int64, device=a.device)
    nn = torch.empty(a.shape[0], dtype=torch.int32, device=a.device)
    L.call("pcc_nn_sorted_x", L.ptr(a), a.shape[0], L.ptr(b), b.shape[0], L.ptr(d2), L.ptr(nn), L.stream())
    return d2, nn


# The reference rounds colours to k/255 (`metric.py:152-153`) and then truncates k/255*255 to uint8 (`metric.py:175`).
# Under IEEE double division that product truncates back to k for every 8-bit level (checked here), so the level IS k.
# (Dividing a GPU tensor by the scalar 255 multiplies by its reciprocal instead, which lands below k for some levels:
# the integer level is used directly rather than re-deriving it through that arithmetic.)
assert all(int(float(k) / 255.0 * 255.0) == k for k in range(256))


def rgb_to_yuv(rgb):
    """`convert_rgb_to_yuv(clip(round(rgb*255)/255, 0, 1))` (`metrics/metric.py:152-153,170-188`): 8-bit levels with
    the reference's truncation, BT.709, chroma + 0.5."""
    c = torch.round(rgb.to(torch.float64) * 255.0).clamp(0, 255).to(torch.float32)
    m = torch.tensor(_YUV, dtype=torch.float32, device=rgb.device)
    yuv = (c @ m.t()) / 255.0
    yuv[:, 1:] += 0.5
    return yuv


def _one_direction(prefix, a_pts, a_rgb, b_pts, b_rgb, resolution):
    d2, nn = nearest(a_pts, b_pts)
    l2 = d2.to(torch.float64) / 3.0                                  # mean over the three axes, `metric.py:121`
    r = {prefix + "mse": float(l2.mean()), prefix + "hausdorff": float(l2.max())}
    for k in ("mse", "hausdorff"):
        v = r[prefix + k]
        r[prefix + "psnr_" + k] = math.inf if v == 0 else 10 * math.log10(resolution ** 2 / v)
    if a_rgb is not None and b_rgb is not None:
        ya, yb = rgb_to_yuv(a_rgb), rgb_to_yuv(b_rgb[nn.long()])
        e = ((ya - yb) ** 2).to(torch.float64).mean(dim=0)
        for i, ch in enumerate("yuv"):
            r[prefix + ch + "_mse"] = float(e[i])
            r[prefix + ch + "_psnr"] = math.inf if e[i] == 0 else 10 * math.log10(1 / float(e[i]))
        m = float(e.mean())
        r[prefix + "yuv_mse"] = m
        r[prefix + "yuv_psnr"] = math.inf if m == 0 else 10 * math.log10(1 / m)
    return r


def pointcloud_metrics(source, reconstruction, resolution=1023):
    """`PointCloudMetric(source, reconstruction, resolution).compute_pointcloud_metrics(drop_duplicates=True)`:
    source / reconstruction are [N, 3] or [N, 6] (xyz then rgb in [0,1]) GPU tensors of voxel coordinates.
    Ties between equidistant neighbours resolve to the smallest canonical row (a KD-tree's choice is unspecified)."""
    sa = torch.as_tensor(source)
    sb = torch.as_tensor(reconstruction)
    a_pts, a_rgb = _canonical(sa[:, :3], sa[:, 3:6] if sa.shape[1] >= 6 else None)
    b_pts, b_rgb = _canonical(sb[:, :3], sb[:, 3:6] if sb.shape[1] >= 6 else None)
    r = {}
    r.update(_one_direction("AB_", a_pts, a_rgb, b_pts, b_rgb, resolution))
    r.update(_one_direction("BA_", b_pts, b_rgb, a_pts, a_rgb, resolution))
    keys = ["mse", "hausdorff", "psnr_mse", "psnr_hausdorff"]
    if a_rgb is not None and b_rgb is not None:
        keys += [c + s for c in "yuv" for s in ("_mse", "_psnr")]
    for k in keys:                                                   # `metric.py:72-83`: min over the two directions
        r["sym_" + k] = min(r["AB_" + k], r["BA_" + k])
    return r


def d1_psnr(a_xyz, b_xyz, resolution=1023):
    """Point-to-point geometry PSNR (`metrics/metric.py:113-119,74`): (A->B, B->A, symmetric = min)."""
    a, _ = _canonical(torch.as_tensor(a_xyz)[:, :3])
    b, _ = _canonical(torch.as_tensor(b_xyz)[:, :3])
    ab = _one_direction("", a, None, b, None, resolution)["psnr_mse"]
    ba = _one_direction("", b, None, a, None, resolution)["psnr_mse"]
    return ab, ba, min(ab, ba)


def count_bits(strings):
    """`utils.count_bits` (`utils.py:30-48`)."""
    return sum(count_bits(s) if isinstance(s, list) else len(s) * 8 for s in strings)
