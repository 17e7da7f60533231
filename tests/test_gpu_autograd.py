"""BASELINE config 4 building blocks: gradients of the sparse convolutions (data / weight / bias) against a plain
PyTorch fp32 reference of the same op (gather + matmul + index_add on the CPU, autograd), tolerance 1e-4."""
import numpy as np
import pytest
import torch

from oracle import coords as co, codec
from tests.util import dev, t, n, cloud_keys, assert_close

pytestmark = pytest.mark.gpu


def _torch_ref(feats, W, b, pairs, n_out, act):
    """out = act(b + sum_k index_add(feats[in_k] @ W[k])) with autograd (CPU)."""
    out = torch.zeros((n_out, W.shape[2]), dtype=torch.float32) + (b if b is not None else 0)
    for k, (i, o) in enumerate(pairs):
        if len(i):
            out = out.index_add(0, torch.from_numpy(o.astype(np.int64)), feats[torch.from_numpy(i.astype(np.int64))] @ W[k])
    pre = out
    if act == "relu":
        out = torch.relu(out)
    elif act == "leaky":
        out = torch.nn.functional.leaky_relu(out, 0.01)
    return out, pre.detach()


def _check(mod, x, out_fn, pairs, n_out, act, W, b, f):
    fr = torch.from_numpy(f).requires_grad_(True)
    Wr = torch.from_numpy(W).requires_grad_(True)
    br = torch.from_numpy(b).requires_grad_(True) if b is not None else None
    ref, pre = _torch_ref(fr, Wr, br, pairs, n_out, act)
    go = np.random.default_rng(7).standard_normal(ref.shape).astype(np.float32)
    if act is not None:
        # The derivative of ReLU / LeakyReLU jumps at 0: an output whose pre-activation sits within fp32 summation noise of
        # 0 (|pre| < 1e-5 on sums of magnitude ~1) may take either sign depending on the summation order, and its upstream
        # gradient then differs by (1 - slope) * go -- a whole-row error of |go| * |W| in every input row it touches.  That
        # is what the round-3 "intermittent" failure of the 192 -> 192 case was (tools/kink_rate.py: one weight draw in
        # ~20 has such an output; the weights were drawn from an unseeded generator).  Such outputs get no upstream gradient.
        go[np.abs(pre.numpy()) < 1e-5] = 0.0
    ref.backward(torch.from_numpy(go))
    got = out_fn()
    assert_close(n(got), ref.detach().numpy(), what="forward")
    got.backward(t(go))
    assert_close(n(x._F.grad), fr.grad.numpy(), what="data gradient")
    wg = n(mod.kernel.grad)
    assert_close(wg.reshape(W.shape), Wr.grad.numpy(), atol=2e-4, rtol=2e-4, what="weight gradient")
    if b is not None:
        assert_close(n(mod.bias.grad), br.grad.numpy(), atol=2e-4, rtol=2e-4, what="bias gradient")


@pytest.mark.parametrize("cin,cout,ks,stride,act", [(16, 32, 3, 1, "relu"), (32, 16, 3, 1, "leaky"), (128, 128, 5, 2, None),
                                                    (32, 1, 3, 1, None), (16, 1, 3, 1, None), (64, 1, 3, 1, "relu"),
                                                    (8, 3, 1, 1, None), (4, 16, 5, 2, None),
                                                    (192, 192, 3, 2, "leaky")])
def test_conv_gradients(cin, cout, ks, stride, act):
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    from unified_point_cloud_compression_amd import lib as L
    keys = cloud_keys(cin + cout + ks, 16, 0.2, 1, batch=2)
    rng = np.random.default_rng(1)
    f = rng.standard_normal((len(keys), cin)).astype(np.float32)
    mod = ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=stride, bias=True, dimension=3).to(dev())
    W = n(mod.kernel).reshape(ks ** 3, cin, cout).copy() * 3
    with torch.no_grad():
        mod.kernel.copy_(t(W.reshape(n(mod.kernel).shape)))
    b = n(mod.bias).copy()
    x = ME.SparseTensor(coordinates=t(co.unpack_keys(keys)), features=t(f).requires_grad_(True))
    out_keys = keys if stride == 1 else co.stride_keys(keys, stride)
    pairs = codec.kernel_map_pairs(keys, out_keys, ks, 1) if ks > 1 else [(np.arange(len(keys)), np.arange(len(keys)))]
    code = {None: L.ACT_NONE, "relu": L.ACT_RELU, "leaky": L.ACT_LEAKY}[act]

    def run():
        cs = x._cset
        out_set = cs if stride == 1 else cs.stride(stride)
        kmap = None if ks == 1 else cs.kernel_map(out_set, ks)
        return mod._apply_conv(x, out_set, kmap, act=code)
    _check(mod, x, run, pairs, len(out_keys), act, W, b, f)


@pytest.mark.parametrize("cin,cout", [(16, 1), (32, 1), (64, 1), (32, 16), (16, 16)])
def test_self_map_weight_gradient_equals_the_pair_list_form(cin, cout):
    """`pcc_conv_wgrad_self` (input-stationary, inverse offsets of the set's own map) against `pcc_conv_wgrad` on the same map
    and against the float64 sum over the oracle's pairs; a set large enough for several workgroup chunks and ragged tails."""
    from unified_point_cloud_compression_amd import lib as L
    from unified_point_cloud_compression_amd import sparse as S
    keys = cloud_keys(cin, 40, 0.12, 1, batch=3)
    rng = np.random.default_rng(cin)
    f = rng.standard_normal((len(keys), cin)).astype(np.float32)
    g = rng.standard_normal((len(keys), cout)).astype(np.float32)
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    cs = ME.SparseTensor(coordinates=t(co.unpack_keys(keys)), features=t(f))._cset      # rows in key order (keys are canonical)
    kmap = cs.kernel_map(cs, 3)
    assert L.load().pcc_conv_wgrad_self_supported(27, cin, cout)
    a = n(S.conv_wgrad_self(t(f), t(g), 27, cin, kmap, cout))
    b = n(S.conv_wgrad(t(f), t(g), 27, cin, cout, kmap))
    ref = np.zeros((27, cin, cout))
    for k, (i, o) in enumerate(codec.kernel_map_pairs(keys, keys, 3, 1)):
        if len(i):
            ref[k] = f[i].astype(np.float64).T @ g[o].astype(np.float64)
    assert_close(a, b, atol=2e-4, rtol=2e-4, what="self form vs pair-list form")
    assert_close(a, ref.astype(np.float32), atol=2e-4, rtol=2e-4, what="self form vs float64 pairs")
    assert not L.load().pcc_conv_wgrad_self_supported(27, cin, 2) and not L.load().pcc_conv_wgrad_self_supported(8, cin, 1)
    assert not L.load().pcc_conv_wgrad_self_supported(27, 64, 16)


@pytest.mark.parametrize("cin,cout,ks", [(16, 16, 5), (128, 32, 5), (32, 32, 2), (192, 192, 2)])
def test_generative_transpose_gradients(cin, cout, ks):
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    keys = cloud_keys(cin + ks, 9, 0.15, 2)
    rng = np.random.default_rng(2)
    f = rng.standard_normal((len(keys), cin)).astype(np.float32)
    mod = ME.MinkowskiGenerativeConvolutionTranspose(cin, cout, kernel_size=ks, stride=2, bias=True, dimension=3).to(dev())
    W = n(mod.kernel).copy() * 3
    with torch.no_grad():
        mod.kernel.copy_(t(W))
    b = n(mod.bias).copy()
    x = ME.SparseTensor(coordinates=t(co.unpack_keys(keys)), features=t(f).requires_grad_(True), tensor_stride=2)
    out_keys = co.expand_keys(keys, ks, 1)
    pairs = codec.kernel_map_pairs(keys, out_keys, ks, 1, transposed=True)
    _check(mod, x, lambda: mod(x).F, pairs, len(out_keys), None, W, b, f)


def test_gaussian_likelihood_kernels_match_torch_autograd():
    """`pcc_gauss_lik_fwd/bwd` (training rate term, `model/entropy_models.py:312-316`, `loss.py:77-79`) against the torch
    formula and its autograd gradients, including elements on both lower bounds (scale < 0.11, likelihood < 1e-9)."""
    from unified_point_cloud_compression_amd.compressai.entropy_models import GaussianConditional
    gc = GaussianConditional(None).to(dev())
    rng = np.random.default_rng(2)
    v = rng.standard_normal((700, 24)).astype(np.float32) * 4
    s = np.exp(rng.uniform(-4, 2, (700, 24))).astype(np.float32)            # scales on both sides of the 0.11 bound
    mu = rng.standard_normal((700, 24)).astype(np.float32)
    v[:40] += 60.0                                                           # far tails: likelihood on the 1e-9 floor
    w = rng.standard_normal((700, 24)).astype(np.float32)
    res = []
    for fused in (True, False):
        tv, ts, tm = (t(a).requires_grad_(True) for a in (v, s, mu))
        if fused:
            lik = gc.likelihood_rows(tv, ts, tm)
        else:
            sb = gc.lower_bound_scale(ts)
            a = torch.abs(tv - tm)
            lik = gc.likelihood_lower_bound(gc._standardized_cumulative((0.5 - a) / sb) - gc._standardized_cumulative((-0.5 - a) / sb))
        (-(torch.log2(lik)) * t(w)).sum().backward()
        res.append((n(lik), n(tv.grad), n(ts.grad), n(tm.grad)))
    for got, want, what in zip(res[0], res[1], ("likelihood", "d/dv", "d/dscale", "d/dmean")):
        assert_close(got, want, atol=1e-5, rtol=1e-4, what=what)
    assert (res[0][0] == np.float32(1e-9)).sum() >= 40 * 24 * 0.9            # the floor really was exercised


def test_factorised_prior_likelihood_kernels_match_the_torch_chain():
    """`pcc_eb_lik_fwd / bwd` (training likelihood of the factorised prior, `model/entropy_models.py:272,282-285`) against
    the element-wise torch chain they replace: likelihoods, d v and the gradient of every raw parameter (matrices through
    softplus, factors through tanh, biases) -- including rows at the 1e-9 floor, where CompressAI's LowerBound rule decides."""
    from unified_point_cloud_compression_amd.compressai.entropy_models import EntropyBottleneck
    torch.manual_seed(3)
    c, rows = 24, 517
    eb = EntropyBottleneck(c).to(dev())
    with torch.no_grad():
        for nme, p in eb.named_parameters():
            if nme != "quantiles":
                p.add_(torch.randn_like(p) * 0.3)
    v0 = torch.randn(rows, c, device=dev()) * 6.0
    v0[::7] *= 40.0                                             # far tails: the likelihood sits on its floor there
    go = torch.randn(rows, c, device=dev())

    def chain(v):
        x = v.t().unsqueeze(1)
        lo, up = eb._logits_cumulative(x - 0.5), eb._logits_cumulative(x + 0.5)
        sg = -torch.sign(lo + up).detach()
        lik = torch.abs(torch.sigmoid(sg * up) - torch.sigmoid(sg * lo))[:, 0, :].t()
        return eb.likelihood_lower_bound(lik)

    res = {}
    for tag, fn in (("kernel", eb.likelihood_rows), ("torch", chain)):
        eb.zero_grad(set_to_none=True)
        v = v0.clone().requires_grad_(True)
        lik = fn(v)
        lik.backward(go)
        res[tag] = (lik.detach(), v.grad.clone(), {nme: p.grad.clone() for nme, p in eb.named_parameters() if p.grad is not None})
    assert (res["kernel"][0] == 1e-9).any()                     # the floor is exercised
    assert_close(n(res["kernel"][0]), n(res["torch"][0]), atol=1e-7, rtol=1e-5, what="likelihood")
    assert_close(n(res["kernel"][1]), n(res["torch"][1]), atol=1e-6, rtol=1e-4, what="d v")
    assert set(res["kernel"][2]) == set(res["torch"][2]) and len(res["torch"][2]) == 14
    for nme, gt in res["torch"][2].items():
        gk = res["kernel"][2][nme]
        assert_close(n(gk), n(gt), atol=1e-5 * max(1.0, float(gt.abs().max())), rtol=2e-4, what=f"gradient of {nme}")


def test_quant_offset_network_kernels_match_the_torch_layers():
    """`pcc_quant_mlp_fwd/bwd` (`quant_nn`, reference `model/entropy_models.py:210-233`) against the same network as torch layers:
    outputs, input gradients and all six parameter gradients (sums over 300 k elements: relative tolerance)."""
    from unified_point_cloud_compression_amd.autograd import QuantMlpFn
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(2, 10), torch.nn.ReLU(), torch.nn.Linear(10, 10), torch.nn.ReLU(),
                              torch.nn.Linear(10, 1)).to(dev())
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(2.0)
    rng = np.random.default_rng(5)
    s0 = rng.uniform(0.2, 3.0, (2345, 128)).astype(np.float32)
    d0 = rng.uniform(0.1, 9.0, (2345, 128)).astype(np.float32)
    go = rng.standard_normal((2345, 128)).astype(np.float32)
    s1, d1 = t(s0).requires_grad_(True), t(d0).requires_grad_(True)
    ref = net(torch.stack([s1, d1], dim=-1)).squeeze(-1)
    ref.backward(t(go))
    ref_grads = [p.grad.clone() for p in net.parameters()]
    ref_ds, ref_dd = s1.grad.clone(), d1.grad.clone()
    for p in net.parameters():
        p.grad = None
    s2, d2 = t(s0).requires_grad_(True), t(d0).requires_grad_(True)
    out = QuantMlpFn.apply(s2, d2, net[0].weight, net[0].bias, net[2].weight, net[2].bias, net[4].weight, net[4].bias)
    assert_close(n(out), n(ref.detach()), atol=1e-5, rtol=1e-5, what="offsets")
    out.backward(t(go))
    assert_close(n(s2.grad), n(ref_ds), atol=1e-5, rtol=1e-5, what="d scale")
    assert_close(n(d2.grad), n(ref_dd), atol=1e-5, rtol=1e-5, what="d stddev")
    for p, r in zip(net.parameters(), ref_grads):
        scale = float(r.abs().max()) + 1e-6
        assert_close(n(p.grad) / scale, n(r) / scale, atol=2e-5, rtol=0, what="parameter gradient")
    # no gradient for the (detached) gain, as the training forward calls it
    d3 = t(d0).requires_grad_(True)
    QuantMlpFn.apply(t(s0), d3, net[0].weight, net[0].bias, net[2].weight, net[2].bias, net[4].weight, net[4].bias).sum().backward()
    assert d3.grad is not None


def test_fused_focal_loss_rows_match_the_torch_chain():
    """`pcc_focal_rows` (one occupancy level of `Multiscale_FocalLoss`, reference `loss.py:115-157`) against the torch operator
    chain: the level's mean and the gradient with respect to the logits, with logits far enough out for the clip to bind."""
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    from unified_point_cloud_compression_amd import loss as LS
    keys = cloud_keys(11, 24, 0.2, 1, batch=3)
    gt_keys = keys[np.random.default_rng(1).random(len(keys)) < 0.4]
    rng = np.random.default_rng(2)
    lg = (rng.standard_normal((len(keys), 1)) * 4).astype(np.float32)
    q_map = t(np.array([[0.3, 1.0], [2.0, 5.0], [0.7, 9.0]], dtype=np.float32))
    gt = ME.SparseTensor(coordinates=t(co.unpack_keys(gt_keys)), features=t(np.ones((len(gt_keys), 1), np.float32)))
    fl = LS.Multiscale_FocalLoss({"id": "f", "alpha": 0.7, "gamma": 2.0})
    vals, grads = [], []
    for fused in (True, False):
        LS.FUSED_FOCAL = fused
        try:
            x = t(lg).requires_grad_(True)
            pred = ME.SparseTensor(coordinates=t(co.unpack_keys(keys)), features=x)
            out = fl(gt, {"occ_predictions": [pred], "points": [gt], "q_map": q_map})
            out.backward()
            vals.append(float(out.detach()))
            grads.append(n(x.grad))
        finally:
            LS.FUSED_FOCAL = True
    assert abs(vals[0] - vals[1]) <= 1e-5 * abs(vals[1])
    assert_close(grads[0] * len(keys), grads[1] * len(keys), atol=1e-5, rtol=1e-4, what="d loss / d logits")


@pytest.mark.parametrize("c,inverse", [(128, False), (128, True), (32, True)])
def test_gdn_gradients_match_torch_autograd_of_the_formula(c, inverse):
    """`GdnFn` (fused forward; backward = library products + the element-wise / reparametrisation kernels of round 4) against
    torch autograd over the GDN1 formula of `model/blocks.py:38-57` on the CPU, and against the torch-operator backward."""
    import unified_point_cloud_compression_amd.autograd as AG
    from unified_point_cloud_compression_amd.model.blocks import MinkowskiGDN
    torch.manual_seed(c + int(inverse))
    m = MinkowskiGDN(c, inverse=inverse).to(dev())
    with torch.no_grad():
        m.gamma.add_(torch.rand_like(m.gamma) * 0.05)
        m.beta.mul_(1.0 + torch.rand_like(m.beta))
        m.gamma[0, :4] = 0.0                                            # entries AT the lower bound: the LowerBound gradient rule
    rng = np.random.default_rng(c)
    x0 = rng.standard_normal((3001, c)).astype(np.float32)
    go = rng.standard_normal((3001, c)).astype(np.float32)
    # reference: torch autograd on the CPU
    mc = MinkowskiGDN(c, inverse=inverse)
    mc.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    xc = torch.from_numpy(x0).requires_grad_(True)
    beta, gamma = mc.beta_reparam(mc.beta), mc.gamma_reparam(mc.gamma)
    nrm = beta + xc.abs() @ gamma.t()
    yc = xc * nrm if inverse else xc / nrm
    yc.backward(torch.from_numpy(go))
    outs = []
    for fused in (True, False):
        AG.GDN_FUSED_BWD = fused
        try:
            for p in m.parameters():
                p.grad = None
            xg = t(x0).requires_grad_(True)
            y = m.forward_rows(xg)
            y.backward(t(go))
            outs.append((n(y.detach()), n(xg.grad), n(m.beta.grad), n(m.gamma.grad)))
        finally:
            AG.GDN_FUSED_BWD = True
    ref = (yc.detach().numpy(), xc.grad.numpy(), mc.beta.grad.numpy(), mc.gamma.grad.numpy())
    for name, a, b, r in zip(("y", "dx", "d beta", "d gamma"), outs[0], outs[1], ref):
        scale = float(np.abs(r).max()) + 1e-12
        assert_close(a / scale, r / scale, atol=2e-5, rtol=0, what=f"fused {name} vs torch autograd")
        assert_close(a / scale, b / scale, atol=2e-5, rtol=0, what=f"fused {name} vs torch-operator backward")
