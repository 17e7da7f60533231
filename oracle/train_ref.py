"""Training-step restatement (BASELINE config 4) in plain PyTorch fp32 on the CPU -- test infrastructure.

Forward of `UnifiedModel.forward` in training mode (`model/model.py:45-90`, `model/entropy_models.py:236-340`,
`model/transforms.py:170-225`) and the losses of `loss.py:63-157`, built from gather + matmul + index_add over the
oracle's kernel maps, so `loss.backward()` gives reference gradients for every parameter.  Noise of the quantisation
proxies is supplied by the caller so both sides use the same draw.
"""
import numpy as np
import torch

from . import coords as co
from . import codec

PEDESTAL = 2.0 ** -36


def _idx(a):
    return torch.from_numpy(np.asarray(a, dtype=np.int64))


class _LowerBound(torch.autograd.Function):
    """CompressAI `LowerBound`: max(x, b); the gradient passes where x >= b OR it would push x up."""

    @staticmethod
    def forward(ctx, x, bound):
        b = torch.tensor(float(bound), dtype=x.dtype)
        ctx.save_for_backward(x, b)
        return torch.max(x, b)

    @staticmethod
    def backward(ctx, g):
        x, b = ctx.saved_tensors
        return ((x >= b) | (g < 0)).to(g.dtype) * g, None


def lower_bound(x, b):
    return _LowerBound.apply(x, b)


def sparse_conv(f, W, b, pairs, n_out):
    W3 = W if W.dim() == 3 else W.unsqueeze(0)
    out = torch.zeros((n_out, W3.shape[2]), dtype=torch.float32)
    if b is not None:
        out = out + b.reshape(1, -1)
    for k, (i, o) in enumerate(pairs):
        if len(i):
            out = out.index_add(0, _idx(o), f[_idx(i)] @ W3[k])
    return out


def conv_layer(P, name, keys, f, ts, ks, stride=1, transposed=False, bias=True):
    W = P[name + ".kernel"]
    b = P.get(name + ".bias") if bias else None
    if transposed:
        ts_out = ts // stride
        out_keys = co.expand_keys(keys, ks, ts_out)
        pairs = codec.kernel_map_pairs(keys, out_keys, ks, ts_out, transposed=True)
    else:
        ts_out = ts * stride
        out_keys = keys if stride == 1 else co.stride_keys(keys, ts_out)
        pairs = [(np.arange(len(keys)), np.arange(len(keys)))] if ks == 1 else codec.kernel_map_pairs(keys, out_keys, ks, ts)
    return out_keys, sparse_conv(f, W, b, pairs, len(out_keys)), ts_out


def reparam(x, minimum=0.0):
    bound = (minimum + PEDESTAL) ** 0.5
    return lower_bound(x, bound) ** 2 - PEDESTAL


def gdn(P, name, f, inverse):
    beta = reparam(P[name + ".beta"], 1e-6)
    gamma = reparam(P[name + ".gamma"])
    norm = f.abs() @ gamma.t() + beta
    return f * norm if inverse else f / norm


def std_cum(x):
    return 0.5 * torch.erfc(-(2 ** -0.5) * x)


def gaussian_likelihood(v, scales, means):
    s = lower_bound(scales, 0.11)
    a = (v - means).abs()
    return lower_bound(std_cum((0.5 - a) / s) - std_cum((-0.5 - a) / s), 1e-9)


def eb_logits(P, x):     # x [C,1,N]
    pre = "entropy_model.entropy_bottleneck."
    for i in range(5):
        x = torch.matmul(torch.nn.functional.softplus(P[pre + f"_matrix{i}"]), x) + P[pre + f"_bias{i}"]
        if i < 4:
            x = x + torch.tanh(P[pre + f"_factor{i}"]) * torch.tanh(x)
    return x


def eb_likelihood(P, v):  # v [N,C]
    x = v.t().unsqueeze(1)
    lo, up = eb_logits(P, x - 0.5), eb_logits(P, x + 0.5)
    sg = -torch.sign(lo + up).detach()
    return lower_bound((torch.sigmoid(sg * up) - torch.sigmoid(sg * lo)).abs(), 1e-9)[:, 0, :].t()


def mlp(P, pre, x, n, softplus=False):
    idx = 0
    for li in range(n):
        x = x @ P[f"{pre}.{idx}.weight"].t() + P[f"{pre}.{idx}.bias"]
        idx += 2
        if li < n - 1:
            x = torch.relu(x)
    return torch.nn.functional.softplus(x) if softplus else x


def forward_loss(P, cfg, C, rgb, q, Lambda, noise_y, noise_z, loss_cfg):
    """P: dict of torch tensors (requires_grad).  C [N,4] int coords (unique), rgb [N,3].  Returns (total, parts)."""
    em = cfg["entropy_model"]
    keys, first = co.canonicalize(C)
    f = torch.cat([torch.ones((len(keys), 1)), torch.from_numpy(np.asarray(rgb, np.float32)[first])], dim=1)
    gt_keys, gt_rgb = keys, f[:, 1:]
    batch = torch.from_numpy((keys >> 48).astype(np.int64))
    # ---- g_a
    k = [codec.count_per_batch(keys)]
    ks_, f_, ts = conv_layer(P, "g_a.down_conv_1.0", keys, f, 1, 5, 2)
    f_ = gdn(P, "g_a.down_conv_1.1", f_, False); k.append(codec.count_per_batch(ks_)); k1_keys = ks_
    ks_, f_, ts = conv_layer(P, "g_a.down_conv_2.0", ks_, f_, ts, 5, 2)
    f_ = gdn(P, "g_a.down_conv_2.1", f_, False); k.append(codec.count_per_batch(ks_)); k2_keys = ks_
    ks_, f_, ts = conv_layer(P, "g_a.down_conv_3.0", ks_, f_, ts, 5, 2)
    f_ = gdn(P, "g_a.down_conv_3.1", f_, False)
    y_keys, y, ts = conv_layer(P, "g_a.down_conv_3.2", ks_, f_, ts, 5, 1)
    k.reverse()
    yb = torch.from_numpy((y_keys >> 48).astype(np.int64))
    # ---- entropy model (training)
    pre = "entropy_model.h_a."
    zk, z, t2 = conv_layer(P, pre + "0", y_keys, y, 8, 3, 1, bias=False); z = torch.nn.functional.leaky_relu(z, 0.01)
    zk, z, t2 = conv_layer(P, pre + "2", zk, z, t2, 3, 2, bias=False); z = torch.nn.functional.leaky_relu(z, 0.01)
    zk, z, t2 = conv_layer(P, pre + "4", zk, z, t2, 3, 2, bias=False)
    if em.get("adaptive_BN", True):
        scale = (mlp(P, "entropy_model.scale_nn", q, 3, True) + 1e-4)[yb]
        rescale = (1.0 / scale.detach()) if em["inverse_rescaling"] else (1.0 / mlp(P, "entropy_model.rescale_nn", q, 3, True))[yb]
    else:
        scale = rescale = torch.ones_like(y)
    med = P["entropy_model.entropy_bottleneck.quantiles"][:, 0, 1].detach()
    if em["quantization_mode"] == "uniform":
        z_hat = z + noise_z
        z_lik = eb_likelihood(P, z_hat)
    else:
        z_lik = eb_likelihood(P, z + noise_z)
        zc = z - med
        z_hat = zc + (torch.round(zc) - zc).detach() + med
    pre = "entropy_model.h_s."
    gk, g, t3 = conv_layer(P, pre + "0", zk, z_hat, 32, 2, 2, transposed=True); g = torch.nn.functional.leaky_relu(g, 0.01)
    gk, g, t3 = conv_layer(P, pre + "2", gk, g, t3, 2, 2, transposed=True); g = torch.nn.functional.leaky_relu(g, 0.01)
    gk, g, t3 = conv_layer(P, pre + "4", gk, g, t3, 3, 1)
    rows = co.lookup(gk, y_keys)
    params = torch.where(_idx(rows >= 0).bool().unsqueeze(1), g[_idx(np.maximum(rows, 0))], torch.zeros(1))
    c = y.shape[1]
    scales_hat, means_hat = params[:, :c], params[:, c:]
    if em["quantization_offset"]:
        tmp = scale * (y - means_hat)
        signs = torch.sign(tmp).detach()
        a = tmp.abs()
        y_q_abs = a + noise_y if em["quantization_mode"] == "uniform" else a + (torch.round(a) - a).detach()
        y_lik = gaussian_likelihood(y * scale + noise_y, scales_hat * scale, means_hat * scale)
        stdev = lower_bound(scales_hat * scale, 0.11)
        off = -mlp(P, "entropy_model.quant_nn", torch.stack([scale.detach(), stdev], dim=-1), 3)[..., 0]
        off = torch.where(y_q_abs < 1e-4, torch.zeros(1), off)
        y_hat = signs * (y_q_abs + off) * rescale + means_hat
    else:
        y_t = y * scale + noise_y
        y_lik = gaussian_likelihood(y_t, scales_hat * scale, means_hat * scale)
        y_hat = y_t * rescale
    # ---- g_s (training: top-k from the encoder's counts, predictions kept for the focal loss)
    from . import ops
    xk, x, ts = conv_layer(P, "g_s.up_1.0", y_keys, y_hat, 8, 5, 1)
    x = gdn(P, "g_s.up_1.1", x, True)
    xk, x, ts = conv_layer(P, "g_s.up_1.2", xk, x, ts, 5, 2, transposed=True)
    preds = []
    for lvl, (up, pred) in enumerate((("g_s.up_1", "g_s.predict_1"), ("g_s.up_2", "g_s.predict_2"), ("g_s.up_3", "g_s.predict_3"))):
        if lvl > 0:
            x = gdn(P, up + ".0", x, True)
            xk, x, ts = conv_layer(P, up + ".1", xk, x, ts, 5, 2, transposed=True)
        _, h, _ = conv_layer(P, pred + ".0", xk, x, ts, 3, 1)
        _, logit, _ = conv_layer(P, pred + ".2", xk, torch.relu(h), ts, 3, 1)
        preds.append((xk, logit))
        mask = ops.topk_mask(logit.detach().numpy()[:, 0], k[lvl], (xk >> 48).astype(np.int64))
        xk, x = xk[mask], x[_idx(np.nonzero(mask)[0])]
    _, col, _ = conv_layer(P, "g_s.color_conv.0", xk, x, ts, 1, 1)
    # ---- losses (`loss.py:63-157`)
    parts = {}
    n_pts = len(gt_keys)
    for name, s in loss_cfg.items():
        if s["type"] == "BPPLoss":
            lik = y_lik if s["key"] == "y" else z_lik
            parts[name] = (torch.log(lik).sum() / (-np.log(2) * n_pts)) * s["weight"]
        elif s["type"] == "ColorLoss":
            rows = co.lookup(xk, gt_keys)
            ov = rows >= 0
            pc = col[_idx(rows[ov])]
            gc = gt_rgb[_idx(np.nonzero(ov)[0])]
            e = (gc - pc) ** 2 if s["loss"] == "L2" else (gc - pc).abs()
            parts[name] = (e * Lambda[batch[_idx(np.nonzero(ov)[0])], 1].unsqueeze(1)).mean()
        elif s["type"] == "Multiscale_FocalLoss":
            tot = 0.0
            for (pk, logit), gk_ in zip(preds[::-1], (gt_keys, k1_keys, k2_keys)):
                occ = torch.from_numpy(co.lookup(gk_, pk) >= 0)
                p = torch.sigmoid(logit[:, 0])
                pt = torch.clamp(torch.where(occ, p, 1 - p), 1e-2, 1)
                al = torch.where(occ, torch.tensor(s["alpha"]), torch.tensor(1 - s["alpha"]))
                fl = -al * (1 - pt) ** s["gamma"] * torch.log(pt)
                tot = tot + (fl * Lambda[_idx((pk >> 48).astype(np.int64)), 0]).mean()
            parts[name] = tot
    return sum(parts.values()), parts
