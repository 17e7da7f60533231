"""BASELINE config 4: one training step (forward + backward + clip + Adam) of the variable-rate model
(`configs/CVPR_inverse_scaling.yaml`: adaptive bottleneck, quantisation offsets, inverse rescaling, STE) on a batch of
cubes, against the plain-PyTorch CPU restatement in oracle/train_ref.py with the same noise draw."""
import copy

import numpy as np
import pytest
import torch

from oracle import codec, train_ref, coords as co
from tests.util import dev, t, n, load_params, assert_close

pytestmark = pytest.mark.gpu

LOSS_CFG = {   # `configs/CVPR_inverse_scaling.yaml:58-75`
    "Multiscale_FocalLoss": {"type": "Multiscale_FocalLoss", "alpha": 0.5, "gamma": 2.0},
    "ColorLoss": {"type": "ColorLoss", "loss": "L2"},
    "bpp-y": {"type": "BPPLoss", "key": "y", "weight": 1.0},
    "bpp-z": {"type": "BPPLoss", "key": "z", "weight": 1.0},
}


def _batch(seed, nb=2, size=24, p=0.1):
    rng = np.random.default_rng(seed)
    Cs, Fs = [], []
    for b in range(nb):
        occ = rng.random((size, size, size)) < p
        xyz = np.argwhere(occ)
        Cs.append(np.concatenate([np.full((len(xyz), 1), b), xyz], axis=1))
        Fs.append(rng.random((len(xyz), 3)).astype(np.float32))
    return np.concatenate(Cs).astype(np.int32), np.concatenate(Fs)


@pytest.mark.parametrize("mode,offsets", [("ste", True), ("uniform", False)])
def test_train_step_matches_torch_reference(mode, offsets):
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    from unified_point_cloud_compression_amd.model import UnifiedModel
    from unified_point_cloud_compression_amd.loss import Loss
    cfg = codec.small_config(adaptive=True, offsets=offsets, inverse=True)
    cfg["entropy_model"]["quantization_mode"] = mode
    Pn = codec.random_params(cfg, 3, gain=4.0)
    C, rgb = _batch(0)
    q = np.array([[0.3, 0.8], [0.3, 0.8]], dtype=np.float32)           # one (q_g, q_a) pair per step (`data/q_func.py:41-42`)
    Lam = np.array([[4.0, 300.0], [4.0, 300.0]], dtype=np.float32)
    rng = np.random.default_rng(5)

    # ---- reference (CPU torch autograd over the oracle's maps)
    P = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in Pn.items()}
    keys, _ = co.canonicalize(C)
    y_keys = co.stride_keys(co.stride_keys(co.stride_keys(keys, 2), 4), 8)
    z_keys = co.stride_keys(co.stride_keys(y_keys, 16), 32)
    noise_y = rng.uniform(-0.5, 0.5, (len(y_keys), cfg["entropy_model"]["C_bottleneck"])).astype(np.float32)
    noise_z = rng.uniform(-0.5, 0.5, (len(z_keys), cfg["entropy_model"]["C_hyper_bottleneck"])).astype(np.float32)
    total_ref, parts_ref = train_ref.forward_loss(P, cfg, C, rgb, torch.from_numpy(q), torch.from_numpy(Lam),
                                                  torch.from_numpy(noise_y), torch.from_numpy(noise_z), LOSS_CFG)
    total_ref.backward()

    # ---- build under test
    mcfg = copy.deepcopy(cfg)
    model = load_params(UnifiedModel(mcfg), Pn).to(dev()).train()
    model.entropy_model.noise_fn = lambda tag, like: t(noise_y if tag.startswith("y") else noise_z)
    x = ME.SparseTensor(coordinates=t(C), features=t(rgb))
    out = model(x, t(q), t(Lam))
    total, parts = Loss(copy.deepcopy(LOSS_CFG))(x, out)
    for name in parts_ref:
        assert abs(float(parts[name].detach()) - float(parts_ref[name].detach())) <= 1e-4 + 1e-4 * abs(float(parts_ref[name].detach())), name
    total.backward()
    sd = dict(model.named_parameters())
    checked = 0
    for name, p_ref in P.items():
        if p_ref.grad is None or name not in sd:
            continue
        g = sd[name].grad
        assert g is not None, name
        gr = p_ref.grad.numpy()
        scale = max(np.abs(gr).max(), 1e-6)
        assert_close(n(g) / scale, gr / scale, atol=2e-4, rtol=2e-4, what=f"grad {name}")
        checked += 1
    assert checked >= 40
    # ---- optimiser step as in `train.py:221-227`: clip to 1.0, Adam
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt = torch.optim.Adam([p for nme, p in model.named_parameters() if not nme.endswith(".quantiles")], lr=1e-4)
    before = sd["g_s.up_2.1.kernel"].detach().clone()
    opt.step()
    assert not torch.equal(before, sd["g_s.up_2.1.kernel"].detach())
    aux = model.aux_loss()                                                  # quantile loss (`train.py:230-234`)
    aux.backward()
    assert sd["entropy_model.entropy_bottleneck.quantiles"].grad is not None


def test_train_step_at_r2_width_matches_the_oracle_fixture():
    """configs[3] at its REAL width (R2 architecture, 4 cubes of 128^3 = 78 288 points, adaptive
    bottleneck + offsets + inverse rescaling + STE) against `tests/golden/train_r2_fixture.npz`, written by
    `tests/golden/make_train_fixture.py` from `oracle/train_ref.py` (CPU torch autograd over the oracle's maps, 28 s): every
    loss part to 1e-4, and for each of the 77 parameter gradients the max-norm, the L2 norm and 64 sampled entries to
    5e-5 of the gradient's max-norm (measured worst case: 5e-6, printed)."""
    import os
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    from unified_point_cloud_compression_amd.model import UnifiedModel
    from unified_point_cloud_compression_amd.loss import Loss
    from tests.golden import make_train_fixture as mk
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_r2_fixture.npz"))
    cfg = mk.train_config()
    C, rgb = mk.batch()
    assert len(C) == int(fx["n_points"])
    nb = int(fx["n_cubes"])
    ny, nz = mk.noise(cfg, C)
    torch.manual_seed(0)
    model = UnifiedModel(copy.deepcopy(cfg)).to(dev()).train()           # the fixture's weights: same seed, same init code
    model.entropy_model.noise_fn = lambda tag, like: t(ny if tag.startswith("y") else nz)
    x = ME.SparseTensor(coordinates=t(C), features=t(rgb))
    q = torch.tensor([[0.4, 0.7]] * nb, device=dev())
    Lam = torch.tensor([[5.0, 400.0]] * nb, device=dev())
    out = model(x, q, Lam)
    total, parts = Loss(copy.deepcopy(mk.LOSS_CFG))(x, out)
    for name, want in zip(fx["part_names"], fx["part_values"]):
        got = float(parts[str(name)].detach())
        assert abs(got - want) <= 1e-4 + 1e-4 * abs(want), (name, got, want)
    assert abs(float(total) - float(fx["total"])) <= 1e-4 * abs(float(fx["total"]))
    total.backward()
    sd = dict(model.named_parameters())
    worst = []
    for i, name in enumerate(fx["grad_names"]):
        name = str(name)
        g = sd[name].grad
        assert g is not None, name
        g = n(g).reshape(-1)
        gmax = float(fx["grad_max"][i])
        idx = mk.sample_idx(name, g.size)
        e_s = float(np.abs(g[idx] - fx["grad_samples"][i][:len(idx)]).max())
        e_m = abs(float(np.abs(g).max()) - gmax)
        e_l = abs(float(np.linalg.norm(g.astype(np.float64))) - float(fx["grad_l2"][i])) / max(float(fx["grad_l2"][i]), 1e-30)
        worst.append((max(e_s, e_m) / max(gmax, 1e-30), e_l, name))
    worst.sort(reverse=True)
    print("worst gradient deviations (|err| / max|grad|, relative L2-norm error, parameter):")
    for w in worst[:8]:
        print(f"   {w[0]:.2e}  {w[1]:.2e}  {w[2]}")
    bad = [w for w in worst if w[0] > 5e-5 or w[1] > 5e-5]
    assert not bad, bad[:5]
    assert len(worst) >= 70
