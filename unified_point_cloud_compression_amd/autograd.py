"""Autograd for the sparse convolutions (BASELINE config 4: one training step forward + backward).

The reference back-propagates through every `ME.MinkowskiConvolution` / `MinkowskiGenerativeConvolutionTranspose`
(`train.py:221-227`).  Here
  * the DATA gradient is the forward kernel again: a convolution through the inverse map with transposed weights
    (for odd kernels the inverse of offset k is offset K-1-k); for the input-stationary transposed conv it is the dense
    GEMM dT @ Wflat^T after scattering grad_out onto the pairs;
  * the WEIGHT gradient is `pcc_conv_wgrad` (MFMA GEMM whose reduction runs over the pair list, deterministic);
  * the bias gradient is a column sum.
Fused activations are differentiated from the saved output.

Round 4 (the training step of BASELINE configs[3] is bound by launch count and host reads as much as by kernels):
  * stride-1 odd kernels over a set mapped onto itself with 1 or 16 output channels take the input-stationary weight gradient
    `pcc_conv_wgrad_self` (feature rows streamed once, only the thin gradient rows gathered);
  * the data gradient's transposed / offset-reversed kernels are packed straight from the parameter (`_pack_view`);
  * `RowSelectFn` (rows a top-k keeps, scatter-free backward), `QuantMlpFn` (`quant_nn`, one kernel per direction),
    `FocalRowsFn` (one occupancy level of the focal loss), `GdnFn.backward` (library products + four element-wise kernels,
    the reparametrisation's gradient included) replace chains of 10-40 torch launches each.
"""
import os

import torch

from . import lib as L
from . import sparse as S


GDN_FUSED_BWD = os.environ.get("PCC_GDN_FUSED_BWD", "1") != "0"  # GDN backward: element-wise parts and reparametrisation as kernels
WGRAD_SELF = os.environ.get("PCC_WGRAD_SELF", "1") != "0"     # one-logit heads: input-stationary weight gradient


def _pack(w3):
    """Pack a [K, cin, cout] weight for conv_forward (one-off tensors of the backward pass)."""
    p = S.PackedConv()
    return p.get(w3.contiguous())


def _pad4(t, dim):
    r = (-t.shape[dim]) % 4
    if r == 0:
        return t
    shape = list(t.shape)
    shape[dim] = r
    return torch.cat([t, t.new_zeros(shape)], dim=dim)


def _pack_view(src3, K, cin, cout, transpose, flip):
    """Packed weights of the convolution W'[k][ci][co] = src3[flip ? K-1-k : k][co][ci] (transpose) straight from the stored
    kernel `src3` -- no flip / permute / copy kernels in front of the pack (`pcc_conv_pack_weights_ex`)."""
    src3 = src3.contiguous()
    n = L.load().pcc_conv_packed_elems(K, cin, cout)
    packed = torch.empty(n, dtype=torch.float32, device=src3.device)
    L.call("pcc_conv_pack_weights_ex", L.ptr(src3), K, cin, cout, 1 if transpose else 0, 1 if flip else 0, L.ptr(packed),
           packed.numel(), L.stream())
    return packed


def _conv_view(feats, src3, transpose, flip, kmap, n_out):
    """The data-gradient convolution with the kernel W' = transposed (and offset-reversed) view of the stored kernel `src3`
    [K, a, b]: W' is [K, b, a].  Shapes the MFMA pack takes directly skip the torch-side view + copy; others go through
    `_conv_any`."""
    K, a, b = src3.shape
    cin, cout = (b, a) if transpose else (a, b)
    if cin % 32 == 0 and cout > 16 and feats.shape[1] == cin:
        return S.conv_forward(feats, _pack_view(src3, K, cin, cout, transpose, flip), None, K, cin, cout, kmap, n_out)
    w = torch.flip(src3, dims=[0]) if flip else src3
    return _conv_any(feats, w.permute(0, 2, 1) if transpose else w, kmap, n_out)


def _conv_any(feats, w3, kmap, n_out):
    """conv_forward for arbitrary (cin, cout): channel counts the kernels do not take are zero-padded to 4 / 32."""
    K, cin, cout = w3.shape
    if cin % 4 != 0:                                   # e.g. grad of a 1- or 3-channel head
        feats, w3 = _pad4(feats, 1), _pad4(w3, 1)
        cin = w3.shape[1]
    if cin > 16 and cin % 32 != 0:                     # MFMA path takes 4/8/16 or multiples of 32
        r = (-cin) % 32
        feats = torch.cat([feats, feats.new_zeros((feats.shape[0], r))], dim=1)
        w3 = torch.cat([w3, w3.new_zeros((K, r, cout))], dim=1)
        cin += r
    if cin not in (4, 8, 16) and cin % 32 != 0:        # 12 -> 16
        r = 16 - cin
        feats = torch.cat([feats, feats.new_zeros((feats.shape[0], r))], dim=1)
        w3 = torch.cat([w3, w3.new_zeros((K, r, cout))], dim=1)
        cin = 16
    return S.conv_forward(feats, _pack(w3), None, K, cin, cout, kmap, n_out)


class SparseConvFn(torch.autograd.Function):
    """out = act(bias + conv(feats; kernel)) over a fixed kernel map (coordinates carry no gradient)."""

    @staticmethod
    def forward(ctx, feats, kernel, bias, module, in_set, out_set, kmap, act, slope):
        K, cin, cout = module.kernel_volume, module.in_channels, module.out_channels
        packed = module._packed.get(kernel, tag_from=module.kernel)
        feats = feats.contiguous()
        if isinstance(kmap, tuple):
            out = S.convt_forward_csr(feats, packed, bias, K, cin, cout, kmap, out_set.n, act, slope)
        elif module.TRANSPOSED:
            out = S.convt_forward(feats, packed, bias, K, cin, cout, kmap, out_set.n, act, slope)
        else:
            out = S.conv_forward(feats, packed, bias, K, cin, cout, kmap, out_set.n, act, slope)
        ctx.save_for_backward(feats, kernel, out if act != L.ACT_NONE else None)
        ctx.meta = (module, in_set, out_set, kmap, act, slope, bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        feats, kernel, out = ctx.saved_tensors
        module, in_set, out_set, kmap, act, slope, has_bias = ctx.meta
        K, cin, cout = module.kernel_volume, module.in_channels, module.out_channels
        g = g.contiguous()
        if act == L.ACT_RELU:
            g = g * (out > 0)
        elif act == L.ACT_LEAKY:
            g = torch.where(out > 0, g, g * slope)
        w3 = kernel.detach() if kernel.dim() == 3 else kernel.detach().unsqueeze(0)        # [K, cin, cout]
        g_feats = g_kernel = g_bias = None
        if has_bias and ctx.needs_input_grad[2]:
            g_bias = g.sum(dim=0, keepdim=True)
        if module.TRANSPOSED:
            if not isinstance(kmap, tuple):
                raise L.PccError("backward of the transposed conv needs the CSR pair lists (sparse.USE_CSR)")
            dT = S.convt_scatter_rows(g, kmap, feats.shape[0] * K, cout).view(feats.shape[0], K * cout)
            if ctx.needs_input_grad[1]:
                gw = S.conv_wgrad(feats, dT, 1, cin, K * cout, None)[0]                      # [cin, K*cout]
                g_kernel = gw.view(cin, K, cout).permute(1, 0, 2).contiguous()
            if ctx.needs_input_grad[0]:
                wt = w3.permute(0, 2, 1).reshape(1, K * cout, cin)                           # Wflat^T
                g_feats = _conv_any(dT, wt, None, feats.shape[0])
        else:
            if ctx.needs_input_grad[1]:
                if (WGRAD_SELF and in_set is out_set and kmap is not None and kmap.rows is None and module.stride == 1
                        and g.shape[0] == feats.shape[0] and L.load().pcc_conv_wgrad_self_supported(K, cin, cout)):
                    g_kernel = S.conv_wgrad_self(feats, g, K, cin, kmap, cout)
                else:
                    g_kernel = S.conv_wgrad(feats, g, K, cin, cout, kmap)
            if ctx.needs_input_grad[0]:
                if K == 1:
                    g_feats = _conv_view(g, w3, True, False, None, feats.shape[0])
                else:
                    if module.kernel_size % 2 == 0:
                        raise L.PccError("backward of even-sized (non-generative) kernels is not supported")
                    inv = out_set.kernel_map(in_set, module.kernel_size, step=in_set.ts)    # roles swapped
                    g_feats = _conv_view(g, w3, True, True, inv, in_set.n)                   # W'[k'] = W[K-1-k']^T
        if g_kernel is not None and kernel.dim() == 2:
            g_kernel = g_kernel[0]
        return g_feats, g_kernel, g_bias, None, None, None, None, None, None


class RowSelectFn(torch.autograd.Function):
    """rows `idx` (unique, ascending: the rows a top-k keeps) of a feature matrix.  The backward pass writes each gradient row
    to its place in a zeroed matrix -- `index_select`'s own backward is an atomic `index_add_` that took 0.38 ms per level
    here, and `f[mask]` counts the mask on the host in both directions."""

    @staticmethod
    def forward(ctx, f, idx):
        ctx.save_for_backward(idx)
        ctx.n = f.shape[0]
        return f.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        out = g.new_zeros((ctx.n, g.shape[1]))
        out.index_put_((idx,), g.contiguous())
        return out, None


class GdnFn(torch.autograd.Function):
    """`MinkowskiGDN.forward` under autograd (reference `model/blocks.py:38-57`, GDN1 form): the forward is the fused MFMA
    kernel `pcc_gdn_fwd`; the backward runs on the library's kernels too --
        n  = beta + |x| gamma^T                          (K = 1 convolution)
        GDN   y = x / n : u = -g y / n,  dx = g / n + sign(x) (u gamma)
        IGDN  y = x n   : u =  g x,      dx = g n   + sign(x) (u gamma)
        dbeta = sum_rows u,   dgamma = u^T |x|              (`pcc_conv_wgrad`, deterministic)
    and the two [C] / [C,C] parameter gradients go through CompressAI's NonNegativeParametrizer by torch autograd."""

    @staticmethod
    def forward(ctx, x, beta_raw, gamma_raw, module):
        x = x.contiguous()
        out = torch.empty_like(x)
        packed, beta_eff = module._pack()
        L.call("pcc_gdn_fwd", L.ptr(x), x.shape[0], module.in_channels, L.ptr(packed), L.ptr(beta_eff),
               1 if module.inverse else 0, L.ptr(out), L.arith(), L.stream())
        ctx.save_for_backward(x, beta_raw, gamma_raw)
        ctx.module = module
        return out

    @staticmethod
    def backward(ctx, g):
        x, beta_raw, gamma_raw = ctx.saved_tensors
        m = ctx.module
        g = g.contiguous()
        c = m.in_channels
        if not (GDN_FUSED_BWD and c % 32 == 0):
            return GdnFn._backward_torch(ctx, g)
        n_rows = x.shape[0]
        packed, beta_eff = m._pack()                                   # the forward's pack: W[ci][co] = gamma_eff[co][ci] (+ planes)
        with torch.no_grad():
            ax = x.abs()
            with L.arith_scope(L.ARITH_BF6):                           # (the GDN pack carries no fp16 planes: six-term form)
                n = S.conv_forward(ax, packed, beta_eff, 1, c, c, None, n_rows)                  # beta + |x| gamma^T
                u, dx = torch.empty_like(x), torch.empty_like(x)
                L.call("pcc_gdn_bwd_pre", L.ptr(x), L.ptr(g), L.ptr(n), x.numel(), 1 if m.inverse else 0, L.ptr(u), L.ptr(dx),
                       L.stream())
                gamma_eff = torch.empty((1, c, c), dtype=torch.float32, device=x.device)
                L.call("pcc_gdn_gamma_eff", L.ptr(gamma_raw.detach().contiguous()), c, L.ptr(gamma_eff), L.stream())
                v = S.conv_forward(u, _pack_view(gamma_eff, 1, c, c, False, False), None, 1, c, c, None, n_rows)   # u gamma
            if ctx.needs_input_grad[0]:
                L.call("pcc_gdn_bwd_post", L.ptr(dx), L.ptr(x), L.ptr(v), x.numel(), L.stream())
            d_beta = u.sum(dim=0)
            d_gamma_t = S.conv_wgrad(ax, u, 1, c, c, None)                                        # [1][ci][co] = sum |x|_ci u_co
            gb, gg = torch.empty_like(beta_raw), torch.empty_like(gamma_raw)
            L.call("pcc_gdn_reparam_bwd", L.ptr(beta_raw.detach().contiguous()), L.ptr(gamma_raw.detach().contiguous()), L.ptr(d_beta),
                   L.ptr(d_gamma_t), c, float(m.beta_min), L.ptr(gb), L.ptr(gg), L.stream())
        return (dx if ctx.needs_input_grad[0] else None), gb, gg, None

    @staticmethod
    def _backward_torch(ctx, g):
        """The same gradients with torch operators around the library's products (channel counts the fused path does not take;
        `PCC_GDN_FUSED_BWD=0`; the reference the fused path is tested against)."""
        x, beta_raw, gamma_raw = ctx.saved_tensors
        m = ctx.module
        with torch.enable_grad():
            b_leaf = beta_raw.detach().requires_grad_(True)
            g_leaf = gamma_raw.detach().requires_grad_(True)
            beta = m.beta_reparam(b_leaf)
            gamma = m.gamma_reparam(g_leaf)
        with torch.no_grad():
            ax = x.abs()
            g3 = gamma.detach().unsqueeze(0).contiguous()                  # [1, co, ci]: norm = |x| gamma^T
            n = _conv_view(ax, g3, True, False, None, x.shape[0]) + beta.detach()
            if m.inverse:
                u = g * x
                dx0 = g * n
            else:
                u = -(g * x) / (n * n)
                dx0 = g / n
            dx = dx0 + torch.sign(x) * _conv_view(u, g3, False, False, None, x.shape[0])
            d_beta = u.sum(dim=0)
            d_gamma = S.conv_wgrad(ax, u.contiguous(), 1, ax.shape[1], u.shape[1], None)[0].t()      # [ci][co] = sum |x|_ci u_co
        gb, gg = torch.autograd.grad([beta, gamma], [b_leaf, g_leaf], [d_beta, d_gamma.contiguous()])
        return (dx if ctx.needs_input_grad[0] else None), gb, gg, None


class GaussLikFn(torch.autograd.Function):
    """Gaussian likelihood of the training forward as one kernel per direction (`pcc_gauss_lik_fwd/bwd`)."""

    @staticmethod
    def forward(ctx, v, scale, mean):
        v, scale, mean = v.contiguous(), scale.contiguous(), mean.contiguous()
        lik = torch.empty_like(v)
        L.call("pcc_gauss_lik_fwd", L.ptr(v), L.ptr(scale), L.ptr(mean), v.numel(), L.ptr(lik), L.stream())
        ctx.save_for_backward(v, scale, mean)
        return lik

    @staticmethod
    def backward(ctx, g):
        v, scale, mean = ctx.saved_tensors
        g = g.contiguous()
        dv = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        ds = torch.empty_like(v) if ctx.needs_input_grad[1] else None
        dm = torch.empty_like(v) if ctx.needs_input_grad[2] else None
        L.call("pcc_gauss_lik_bwd", L.ptr(v), L.ptr(scale), L.ptr(mean), L.ptr(g), v.numel(), L.ptr(dv), L.ptr(ds), L.ptr(dm),
               L.stream())
        return dv, ds, dm


class QuantMlpFn(torch.autograd.Function):
    """`quant_nn` (2 -> 10 -> 10 -> 1, ReLU) on per-element (scale, stddev) pairs as one kernel per direction
    (`pcc_quant_mlp_fwd/bwd`; reference `model/entropy_models.py:210-233`).  Parameters in torch.nn.Linear layouts."""

    @staticmethod
    def forward(ctx, scale, stddev, w1, b1, w2, b2, w3, b3):
        scale, stddev = scale.contiguous(), stddev.contiguous()
        params = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in (w1, b1, w2, b2, w3, b3)]).contiguous()
        out = torch.empty_like(stddev)
        L.call("pcc_quant_mlp_fwd", L.ptr(scale), L.ptr(stddev), stddev.numel(), L.ptr(params), L.ptr(out), L.stream())
        ctx.save_for_backward(scale, stddev, params)
        ctx.shapes = [t.shape for t in (w1, b1, w2, b2, w3, b3)]
        return out

    @staticmethod
    def backward(ctx, g):
        scale, stddev, params = ctx.saved_tensors
        g = g.contiguous()
        n = stddev.numel()
        ds = torch.empty_like(scale) if ctx.needs_input_grad[0] else None
        dd = torch.empty_like(stddev) if ctx.needs_input_grad[1] else None
        dp = torch.empty_like(params)
        ws = L.workspace(L.load().pcc_quant_mlp_ws_bytes(n), g.device)
        L.call("pcc_quant_mlp_bwd", L.ptr(scale), L.ptr(stddev), L.ptr(g), n, L.ptr(params), L.ptr(ds), L.ptr(dd), L.ptr(dp),
               L.ptr(ws), ws.numel(), L.stream())
        grads, at = [], 0
        for shp in ctx.shapes:
            k = 1
            for d in shp:
                k *= d
            grads.append(dp[at:at + k].view(shp))
            at += k
        return (ds, dd, *grads)


class FocalRowsFn(torch.autograd.Function):
    """Sum over the rows of one occupancy level of the focal loss terms (`pcc_focal_rows`; reference `loss.py:115-157`):
    forward = one kernel + one sum, backward = one multiply (the derivative with respect to the logits is produced by the
    forward kernel)."""

    @staticmethod
    def forward(ctx, logits, occ_row, keys, q_map, alpha, gamma):       # logits: 1-D (a column view is fine)
        n = logits.shape[0]
        f = torch.empty(n, dtype=torch.float32, device=logits.device)
        df = torch.empty(n, dtype=torch.float32, device=logits.device)
        q_map = q_map.to(torch.float32).contiguous()
        L.call("pcc_focal_rows", logits.data_ptr(), logits.stride(0), L.ptr(occ_row), L.ptr(keys), n, L.ptr(q_map), q_map.stride(0),
               float(alpha), float(gamma), L.ptr(f), L.ptr(df), L.stream())
        ctx.save_for_backward(df)
        return f.sum()

    @staticmethod
    def backward(ctx, g):
        (df,) = ctx.saved_tensors
        return df * g, None, None, None, None, None


class EbLikFn(torch.autograd.Function):
    """Factorised-prior likelihood of the training forward as one kernel per direction (`pcc_eb_lik_fwd/bwd`): values [N, C],
    packed parameters [C, 58] (softplus(matrices) | biases | tanh(factors), built under autograd by the caller)."""

    @staticmethod
    def forward(ctx, v, packed):
        v, packed = v.contiguous(), packed.contiguous()
        lik = torch.empty_like(v)
        L.call("pcc_eb_lik_fwd", L.ptr(v), v.shape[0], v.shape[1], L.ptr(packed), L.ptr(lik), L.stream())
        ctx.save_for_backward(v, packed)
        return lik

    @staticmethod
    def backward(ctx, g):
        v, packed = ctx.saved_tensors
        g = g.contiguous()
        dv = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        dp = torch.empty_like(packed)
        L.call("pcc_eb_lik_bwd", L.ptr(v), L.ptr(g), v.shape[0], v.shape[1], L.ptr(packed), L.ptr(dv), L.ptr(dp), L.stream())
        return dv, (dp if ctx.needs_input_grad[1] else None)
