"""Shared helpers of the test-suite (oracle <-> HIP glue).  The oracle is the checker only."""
import numpy as np
import torch

from oracle import codec, coords as co

ATOL = 1e-4   # BASELINE.json north_star: fp32 features within 1e-4
RTOL = 1e-4


def dev():
    return torch.device("cuda:0")


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return x.to(dtype) if dtype is not None else x


def n(x):
    return x.detach().cpu().numpy()


def cloud_keys(seed, size, p, ts=1, margin=0, batch=1):
    """Random canonical key set on a stride-ts lattice (numpy)."""
    rng = np.random.default_rng(seed)
    Cs = []
    for b in range(batch):
        occ = rng.random((size, size, size)) < p
        xyz = (np.argwhere(occ) + margin) * ts
        Cs.append(np.concatenate([np.full((len(xyz), 1), b, dtype=np.int64), xyz], axis=1))
    C = np.concatenate(Cs, axis=0)
    keys, _ = co.canonicalize(C)
    return keys


def load_params(model, P):
    """Copy an oracle parameter dict into a UnifiedModel (same state-dict names)."""
    sd = model.state_dict()
    for k, v in P.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
        sd[k] = torch.from_numpy(np.ascontiguousarray(v))
    model.load_state_dict(sd)
    return model


def assert_close(got, want, atol=ATOL, rtol=RTOL, what=""):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if got.size == 0:
        return
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    rows = np.unique(np.nonzero(bad)[0])[:16].tolist() if bad.ndim >= 1 and bad.any() else []
    assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} out of tolerance, max err {err.max():.3e} at "
                           f"{np.unravel_index(err.argmax(), err.shape)}; first rows with bad entries {rows}")
