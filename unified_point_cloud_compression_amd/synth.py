"""Seeded synthetic voxel clouds standing in for the datasets the reference evaluates on.

No dataset ships with the reference (`data/.gitignore:5,8,11-13`) and none can be fetched,
so the BASELINE.json configs are realised as:

* `random_block`   -- config 1: 64^3 Bernoulli(p) occupancy, rgb ~ U[0,1)
* `surface_cloud`  -- config 2: voxelised union of 2-manifolds (sphere + torus) in a
                      2^bits grid, ~0.8 M occupied voxels at bits=10, smooth colours + noise
                      (shaped like 8iVFBv2 longdress vox10, `evaluate.py:39-46`)

Both return a float32 [N,6] array: xyz voxel coordinates then rgb in [0,1], i.e. the
`source` tensor `utils.compress_model_ours` builds (`utils.py:438-441`).  numpy only.
"""
import numpy as np


def random_block(seed=0, size=64, p=0.05):
    rng = np.random.default_rng(seed)
    occ = rng.random((size, size, size)) < p
    xyz = np.argwhere(occ).astype(np.float32)
    rgb = rng.random((xyz.shape[0], 3), dtype=np.float32)
    perm = rng.permutation(xyz.shape[0])
    return np.concatenate([xyz, rgb], axis=1)[perm]


def _voxelise(pts, grid):
    v = np.floor(pts).astype(np.int64)
    v = v[np.all((v >= 0) & (v < grid), axis=1)]
    key = (v[:, 0] << 40) | (v[:, 1] << 20) | v[:, 2]
    _, first = np.unique(key, return_index=True)
    return v[first]


def surface_cloud(seed=0, bits=10, scale=1.0, shuffle=True):
    """Sphere (R = 0.19 grid) + torus surface, voxelised.  bits=10, scale=1 gives ~0.8 M voxels."""
    rng = np.random.default_rng(seed)
    grid = 1 << bits
    c = grid / 2.0
    R = 0.19 * grid * scale
    # sample each surface densely enough that every crossed voxel is hit (2 samples / voxel edge)
    n_s = int(4 * np.pi * R * R * 6)
    u = rng.random(n_s)
    v = rng.random(n_s)
    th = 2 * np.pi * u
    ph = np.arccos(2 * v - 1)
    sph = np.stack([np.sin(ph) * np.cos(th), np.sin(ph) * np.sin(th), np.cos(ph)], axis=1) * R
    sph += c + rng.normal(0, 0.02, 3) * grid * 0.1
    Rt, rt = 0.28 * grid * scale, 0.015 * grid * scale
    n_t = int(4 * np.pi ** 2 * Rt * rt * 6)
    a = 2 * np.pi * rng.random(n_t)
    b = 2 * np.pi * rng.random(n_t)
    tor = np.stack([(Rt + rt * np.cos(b)) * np.cos(a), (Rt + rt * np.cos(b)) * np.sin(a), rt * np.sin(b)],
                   axis=1)
    tor += np.array([c, c, c * (0.9 + 0.1 * rng.random())])
    vox = _voxelise(np.concatenate([sph, tor], axis=0), grid)
    p = vox / float(grid)
    rgb = np.stack([0.5 + 0.5 * np.sin(6.0 * p[:, 0] + 1.0), 0.5 + 0.5 * np.cos(5.0 * p[:, 1]),
                    0.5 + 0.5 * np.sin(4.0 * (p[:, 2] + p[:, 0]))], axis=1)
    rgb = np.clip(rgb + rng.uniform(-0.05, 0.05, rgb.shape), 0.0, 1.0)
    out = np.concatenate([vox.astype(np.float32), rgb.astype(np.float32)], axis=1)
    if shuffle:
        out = out[rng.permutation(out.shape[0])]
    return out
