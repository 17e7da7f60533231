"""Analysis / synthesis transforms g_a, g_s: counterpart of the reference's `model/transforms.py`.

Module structure, attribute names and Sequential indices equal the reference's (`model/transforms.py:32-44,
126-166`) so its `state_dict` keys load.  The forward passes produce the same tensors but
  * fuse ReLU into the first occupancy-head convolution's epilogue,
  * select the top-k rows with a radix select on the device and compact directly, instead of
    torch.topk + int64 flattening + torch.isin + MinkowskiPruning (`model/transforms.py:228-282`),
  * obtain z / training target coordinates with the coordinate-only stride operator,
  * (inference) evaluate an up-sampling block together with its occupancy head: the generative transposed convolution
    and the head's first convolution are both affine, so head_conv(genT(x)) is ONE generative convolution from the
    parents with the composite 7x7x7 kernel M[d] = sum_{off_k - off_j = d} W_k V_j -- 5-6x fewer FLOPs than convolving
    all ~40 candidates per parent with the 128-channel up-sampled features; the up-sampled features themselves are then
    computed only for the rows the top-k keeps (`SparseSynthesisTransform.FUSE_UP_PREDICT`).
"""
import os

import torch
import torch.nn as nn

from .. import MinkowskiEngine as ME
from .. import lib as L
from .. import sparse as S
from ..MinkowskiEngine.sparse_tensor import SparseTensor
from .blocks import MinkowskiGDN


def batch_segments(cset):
    """Row ranges per batch index of a canonical set: ([begin_0, ..., end], batch ids).  Cached on the set."""
    return batch_segments_many([cset])[0]


def batch_segments_begin(cs):
    """Queue the batch ranges of a set whose size the host knows (`pcc_batch_bounds`); returns a Pending for `S.resolve` (the
    caller reads it together with whatever else it waits for), or None when there is nothing to read."""
    if cs.bounds.bmax == 0 or "segments" in cs._derived or cs.bounds.bmax > 10 or cs.n == 0 or not S.BATCH_BOUNDS:
        return None
    entries = cs.bounds.bmax + 2
    a, b, c = L.counter(4), L.counter(4), L.counter(4)
    L.call("pcc_batch_bounds", L.ptr(cs.keys), None, cs.n, entries, L.cptr(a), L.cptr(b), L.cptr(c), L.stream())

    def finish(v):
        cs._derived["segments"] = [int(x) for x in (list(v[0]) + list(v[1]) + list(v[2]))[:entries]]
        return cs._derived["segments"]
    return S.Pending([a, b, c], finish)


def batch_segments_many(csets):
    """`batch_segments` of several sets with ONE device->host read for all of them (the analysis transform knows its three
    sets before its first feature kernel)."""
    todo = [cs for cs in csets if cs.bounds.bmax > 0 and "segments" not in cs._derived]
    if todo:
        pos = [torch.searchsorted(cs.keys[:cs.n], torch.arange(0, cs.bounds.bmax + 2, device=cs.device, dtype=torch.int64) << 48)
               for cs in todo]
        flat = (torch.cat(pos) if len(pos) > 1 else pos[0]).tolist()
        at = 0
        for cs in todo:
            m = cs.bounds.bmax + 2
            cs._derived["segments"] = flat[at:at + m]
            at += m
    return [([0, cs.n], [0]) if cs.bounds.bmax == 0 else (cs._derived["segments"], list(range(cs.bounds.bmax + 1)))
            for cs in csets]


def count_per_batch(x):
    """`AnalysisTransform.count_per_batch` (`model/transforms.py:47-64`): rows per occupied batch index."""
    seg, _ = batch_segments(x._cset)
    return [seg[i + 1] - seg[i] for i in range(len(seg) - 1) if seg[i + 1] > seg[i]]


class AnalysisTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        C_in, N1, N2, N3, N4 = (config[k] for k in ("C_in", "N1", "N2", "N3", "N4"))
        self.down_conv_1 = nn.Sequential(
            ME.MinkowskiConvolution(in_channels=C_in, out_channels=N1, kernel_size=5, stride=2, bias=True, dimension=3),
            MinkowskiGDN(N1),
        )
        self.down_conv_2 = nn.Sequential(
            ME.MinkowskiConvolution(in_channels=N1, out_channels=N2, kernel_size=5, stride=2, bias=True, dimension=3),
            MinkowskiGDN(N2),
        )
        self.down_conv_3 = nn.Sequential(
            ME.MinkowskiConvolution(in_channels=N2, out_channels=N3, kernel_size=5, stride=2, bias=True, dimension=3),
            MinkowskiGDN(N3),
            ME.MinkowskiConvolution(in_channels=N3, out_channels=N4, kernel_size=5, stride=1, bias=True, dimension=3),
        )

    count_per_batch = staticmethod(count_per_batch)

    def plan(self, cs, backward=False, extra=()):
        """Coordinate-only pre-pass: every output set, kernel map and pair list of the transform depends on the
        input COORDINATES alone, so they are all queued before the first feature kernel and the sizes the host needs (pair
        counts of the three 5x5x5 128-channel layers) come back in ONE read instead of one per layer."""
        pend = []
        first = True
        for seq in (self.down_conv_1, self.down_conv_2, self.down_conv_3):
            for m in seq:
                if isinstance(m, ME.MinkowskiConvolution):
                    out = cs if m.stride == 1 else cs.stride(cs.ts * m.stride)   # (queued by the input set's stride chain)
                    if m.kernel_volume > 1 or m.stride != 1:
                        kmap = cs.kernel_map(out, m.kernel_size)
                        if S.wants_pairs(m.kernel_volume, m.in_channels, m.out_channels):
                            pend.append(kmap.pair_plan_begin())
                        # training: the data gradient of a strided layer runs the forward kernel over the INVERSE map
                        # (`autograd.SparseConvFn.backward`); its pair plan is queued here so that its size comes back with
                        # the forward plans' instead of stopping the backward pass for a read of its own
                        if (backward and not first and m.kernel_size % 2 == 1
                                and S.wants_pairs(m.kernel_volume, m.out_channels, m.in_channels)):
                            pend.append(out.kernel_map(cs, m.kernel_size, step=cs.ts).pair_plan_begin())
                    cs = out
                    first = False
        S.resolve(*pend, *[p for p in extra if p is not None])

    def forward(self, x):
        """x -> (y, k) with k = rows per batch at strides [4, 2, 1] (`model/transforms.py:68-97`)."""
        extra = []
        if x._cset.bounds.bmax > 0:                  # batched input (training): the three sets' row ranges, read with the plans
            c1 = x._cset.stride(x._cset.ts * 2)
            extra = [batch_segments_begin(c) for c in (x._cset, c1, c1.stride(c1.ts * 2))]
        self.plan(x._cset, backward=torch.is_grad_enabled(), extra=extra)
        k = [count_per_batch(x)]
        x = self.down_conv_1(x)
        k.append(count_per_batch(x))
        x = self.down_conv_2(x)
        k.append(count_per_batch(x))
        x = self.down_conv_3(x)
        k.reverse()
        return x, k


class SparseSynthesisTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        C_out, N1, N2, N3, N4 = (config[k] for k in ("C_out", "N1", "N2", "N3", "N4"))
        conv, gen = ME.MinkowskiConvolution, ME.MinkowskiGenerativeConvolutionTranspose
        self.up_1 = nn.Sequential(
            conv(in_channels=N4, out_channels=N3, kernel_size=5, stride=1, bias=True, dimension=3),
            MinkowskiGDN(N3, inverse=True),
            gen(in_channels=N3, out_channels=N2, kernel_size=5, stride=2, bias=True, dimension=3),
        )
        self.up_2 = nn.Sequential(
            MinkowskiGDN(N2, inverse=True),
            gen(in_channels=N2, out_channels=N1, kernel_size=5, stride=2, bias=True, dimension=3),
        )
        self.up_3 = nn.Sequential(
            MinkowskiGDN(N1, inverse=True),
            gen(in_channels=N1, out_channels=N1 // 4, kernel_size=5, stride=2, bias=True, dimension=3),
        )
        self.color_conv = nn.Sequential(
            conv(in_channels=N1 // 4, out_channels=C_out, kernel_size=1, stride=1, bias=True, dimension=3),
        )
        self.predict_1 = nn.Sequential(
            conv(in_channels=N2, out_channels=N2 // 2, kernel_size=3, stride=1, bias=True, dimension=3),
            ME.MinkowskiReLU(inplace=False),
            conv(in_channels=N2 // 2, out_channels=1, kernel_size=3, stride=1, bias=True, dimension=3),
        )
        self.predict_2 = nn.Sequential(
            conv(in_channels=N1, out_channels=N1 // 2, kernel_size=3, stride=1, bias=True, dimension=3),
            ME.MinkowskiReLU(inplace=False),
            conv(in_channels=N1 // 2, out_channels=1, kernel_size=3, stride=1, bias=True, dimension=3),
        )
        self.predict_3 = nn.Sequential(
            conv(in_channels=N1 // 4, out_channels=N4 // 8, kernel_size=3, stride=1, bias=True, dimension=3),
            ME.MinkowskiReLU(inplace=False),
            conv(in_channels=N4 // 8, out_channels=1, kernel_size=3, stride=1, bias=True, dimension=3),
        )
        self.prune = ME.MinkowskiPruning()
        self.down_conv = conv(in_channels=1, out_channels=1, kernel_size=3, stride=2, dimension=3)

    # ---- fused building blocks ---------------------------------------------------------------------
    @staticmethod
    def _predict(head, x):
        """conv k3 -> ReLU (fused into the epilogue) -> conv k3 -> 1 logit per row; both convs share one map."""
        c0, c2 = head[0], head[2]
        cs = x._cset
        kmap = cs.kernel_map(cs, 3)
        f = x._canonical_features()
        if (S.HEAD_FUSED and not (torch.is_grad_enabled() and (f.requires_grad or c0.kernel.requires_grad))
                and c0.kernel_size == 3 and c2.kernel_size == 3 and c0.stride == 1 and c2.stride == 1 and c2.out_channels == 1
                and kmap.rows is None and L.load().pcc_conv_head_supported(c0.in_channels, c0.out_channels)):
            # narrow heads (predict_3: 32 -> 16 -> 1): one pass over the features, the hidden layer stays on chip
            w0 = c0._packed.get(c0.kernel, state_dict_order=True)
            w2 = c2._packed.get(c2.kernel, state_dict_order=True)
            logit = S.conv_head_forward(f, w0, c0.bias, c0.out_channels, w2, c2.bias, cs, kmap)
            return SparseTensor._from_canonical(cs, logit)
        h = c0._apply_conv(x, cs, kmap, act=L.ACT_RELU)
        ht = SparseTensor._from_canonical(cs, h)
        logit = c2._apply_conv(ht, cs, kmap)
        return SparseTensor._from_canonical(cs, logit)

    @staticmethod
    def _topk_prediction(prediction, k):
        """Mask of the k[b] largest logits per batch b; ties -> lowest canonical row (SURVEY A.7;
        reference `model/transforms.py:228-254`)."""
        seg, bids = batch_segments(prediction._cset)
        ks = [int(k[b]) if (seg[i + 1] > seg[i]) else 0 for i, b in enumerate(bids)]
        return S.topk_mask(prediction._canonical_features().detach(), seg, ks), sum(
            min(kk, seg[i + 1] - seg[i]) for i, kk in enumerate(ks))

    @staticmethod
    def _prune_tensor(x, mask, n_keep):
        cs = x._cset
        f = x._canonical_features()
        if torch.is_grad_enabled() and f.requires_grad:      # training: row selection through autograd
            # the kept row numbers come from the same stable compaction as the keys (payload = 0..n-1): `f[mask]` would make
            # torch count the mask on the host in the forward AND in the backward pass (two device->host waits per level)
            idx, _, n = S.prune(torch.arange(cs.n, dtype=torch.int64, device=f.device), cs.n, None, mask, n_keep)
            keys = cs.keys.index_select(0, idx)
            from ..autograd import RowSelectFn
            return SparseTensor._from_canonical(S.CoordSet(keys, n, cs.ts, cs.bounds), RowSelectFn.apply(f, idx))
        keys, feats, n = S.prune(cs.keys, cs.n, f, mask, n_keep)
        return SparseTensor._from_canonical(S.CoordSet(keys, n, cs.ts, cs.bounds), feats)

    # ---- up-sampling block + occupancy head as one composite generative convolution (inference) ------------
    FUSE_UP_PREDICT = True
    # all three heads are fused since round 2 (round 1 left predict_3, 32 -> 16, layer-wise: with the fp32 GEMM its per-pair
    # buffer and the 7-wide probing cost what the narrower convolution saved; with the split-path GEMM it is 0.9 ms ahead)
    FUSE_MIN_HEAD_CHANNELS = int(os.environ.get("PCC_FUSE_MIN_HEAD", "16"))

    def _fused_weights(self, gen, c0):
        """(packed composite kernel [343, Cin, Ch], neighbour-existence bias [27, Ch]) of head_conv0(genT(.)), cached per
        parameter version.  Offsets: genT writes parent + off_k, the head reads row + off_j, so parent -> row
        displacement is off_k - off_j, index (ix - jx + 2) per axis in the 7-wide composite."""
        tag = tuple((p.data_ptr(), p._version) for p in (gen.kernel, gen.bias, c0.kernel)) + (S.WEIGHT_OFFSET_ORDER, S.T_Z_FASTEST)
        cache = self.__dict__.setdefault("_fused_cache", {})
        hit = cache.get(id(gen))
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                W, V = gen.kernel.detach().double(), c0.kernel.detach().double()
                pw, pv = S.weight_offset_perm(125, W.device), S.weight_offset_perm(27, V.device)
                W = W if pw is None else W[pw]
                V = V if pv is None else V[pv]
                cin, cm, ch = W.shape[1], W.shape[2], V.shape[2]
                W5, V3 = W.view(5, 5, 5, cin, cm), V.view(3, 3, 3, cm, ch)          # [z][y][x] (x fastest)
                M = torch.zeros((7, 7, 7, cin, ch), dtype=torch.float64, device=W.device)
                for jz in range(3):
                    for jy in range(3):
                        for jx in range(3):
                            M[2 - jz:7 - jz, 2 - jy:7 - jy, 2 - jx:7 - jx] += W5 @ V3[jz, jy, jx]
                if S.T_Z_FASTEST:                                    # offsets numbered z fastest, to match csr_for(zk=True)
                    M = M.permute(2, 1, 0, 3, 4).contiguous()
                Mf = torch.nn.Parameter(M.view(343, cin, ch).float(), requires_grad=False)
                packed = S.PackedConv(transposed=True).get(Mf)
                cb = (gen.bias.detach().double().reshape(1, cm) @ V.reshape(27 * cm, ch).view(27, cm, ch)).reshape(27, ch)
                cache[id(gen)] = hit = (tag, packed, cb.float().contiguous())
        return hit[1], hit[2]

    # A narrow head (predict_3: 32 -> 16) pays for the composite form only when the candidate set is large relative to its
    # parents: the composite GEMM costs n_in * 343 * Cin * Ch whatever the candidates, the layer-wise head n_out * 27 * Cm * Ch.
    # Random weights scatter the kept voxels (66 candidates per parent at the last level: composite 0.9 ms ahead), a trained
    # model keeps them on the surface (21 per parent: layer-wise 1.2 ms ahead) -- measured both ways, round 2.
    FUSE_NARROW_MIN_RATIO = 40

    def _can_fuse(self, up, head, x):
        gen, c0, c2 = up[-1], head[0], head[2]
        ok = (self.FUSE_UP_PREDICT and not torch.is_grad_enabled() and S.USE_GRID and S.USE_CSR and S.EXPAND_BY_GRID
              and isinstance(gen, ME.MinkowskiGenerativeConvolutionTranspose) and gen.kernel_size == 5
              and gen.stride == 2 and gen.bias is not None and c0.kernel_size == 3 and c0.stride == 1
              and c0.out_channels >= self.FUSE_MIN_HEAD_CHANNELS and c0.out_channels % 4 == 0
              and x._cset.n > 0 and x._cset.n * 343 < (1 << 31) and x._cset.grid() is not None)
        if ok and c0.out_channels < 32:
            cs = x._cset
            out_set = cs.expand(5, cs.ts // 2, want_csr=False)       # cached on the set: whichever path runs re-uses it
            ok = out_set.n >= self.FUSE_NARROW_MIN_RATIO * cs.n
        return ok

    def plan(self, cs):
        """Coordinate-only work of the first level that depends on y's coordinates alone: the 5x5x5 map + pair list of
        `up_1[0]` and the candidate set of `up_1`'s generative convolution.  Returns Pendings for the caller to resolve
        together with whatever else it is waiting for (`UnifiedModel.decompress`: the hyper-latent's coordinate set)."""
        pend = []
        for m in list(self.up_1)[:-1]:
            if isinstance(m, ME.MinkowskiConvolution) and m.stride == 1 and m.kernel_volume > 1:
                kmap = cs.kernel_map(cs, m.kernel_size)
                if S.wants_pairs(m.kernel_volume, m.in_channels, m.out_channels):
                    pend.append(kmap.pair_plan_begin())
        gen = self.up_1[-1]
        if isinstance(gen, ME.MinkowskiGenerativeConvolutionTranspose) and cs.ts % gen.stride == 0 and cs.n > 0:
            pend.append(cs.expand_begin(gen.kernel_size, cs.ts // gen.stride, want_csr=False))
        return pend

    def _up_predict_fused(self, up, head, x, k_lvl, probe=None, lvl=0, next_up=None):
        """Returns (x pruned to the top-k rows, prediction over all candidate rows, mask).
        next_up: the following level's up-sampling block -- its candidate set derives from the rows kept here, so that
        expansion is queued with this level's last coordinate work and both sizes come back in one read."""
        for m in list(up)[:-1]:
            x = m(x)
        gen, c0, c2 = up[-1], head[0], head[2]
        cs_in = x._cset
        ts_out = cs_in.ts // 2
        feats = x._canonical_features()
        out_set = cs_in.expand(5, ts_out, want_csr=False)
        packedM, cb = self._fused_weights(gen, c0)
        from_grid = (S.STENCIL_FROM_GRID and out_set.grid() is not None and c2.kernel_size == 3 and c2.stride == 1
                     and c2.out_channels <= 4 and c0.out_channels in (4, 8, 16, 32, 64)
                     and 27 * c2.out_channels * c0.out_channels * 4 <= 48 * 1024)
        chunked = S.T_CHUNKED and cs_in.n * 343 * c0.out_channels * 4 >= S.T_CHUNKED_MIN_BYTES
        # (the one-pass slotted lists are consumed by the grid form of the gather-sum only)
        csr7 = cs_in.csr_for(out_set.keys, out_set.n, 7, ts_out, zk=S.T_Z_FASTEST, slots=S.CSR_SLOTS and from_grid and not chunked)
        if from_grid:
            # the candidate set's own bitmap + rank give the 27 neighbours of a row directly: no 3x3x3 kernel map of the
            # (large) candidate set is built, written and re-read for the presence flags and for the 1-channel convolution
            if chunked:
                h = S.convt_forward_csr_chunked(feats, packedM, c0.bias, 343, gen.in_channels, c0.out_channels, csr7, cs_in,
                                                out_set, L.ACT_RELU, cb)
            else:
                h = S.convt_forward_csr_grid(feats, packedM, c0.bias, 343, gen.in_channels, c0.out_channels, csr7, out_set,
                                             L.ACT_RELU, cb)
            w2 = c2._packed.get(c2.kernel, state_dict_order=True)
            logit = S.conv_thin_grid_forward(h, w2, c2.bias, c0.out_channels, c2.out_channels, out_set)
        else:
            kmap3 = out_set.kernel_map(out_set, 3)
            h = S.convt_forward_csr(feats, packedM, c0.bias, 343, gen.in_channels, c0.out_channels, csr7, out_set.n,
                                    act=L.ACT_RELU, ex_map=kmap3, ex_bias=cb)
            logit = c2._apply_conv(SparseTensor._from_canonical(out_set, h), out_set, kmap3)
        pred = SparseTensor._from_canonical(out_set, logit)
        if probe is None:              # selection and key compaction in one pass over the logits
            seg, bids = batch_segments(out_set)
            ks = [int(k_lvl[b]) if (seg[i + 1] > seg[i]) else 0 for i, b in enumerate(bids)]
            mask, keys, n = S.topk_prune_keys(logit, seg, ks, out_set.keys)
        else:
            mask, n_keep = self._topk_prediction(pred, k_lvl)
            forced = probe("select", lvl, out_set, logit, mask, None)
            if forced is not None:
                mask, n_keep = forced, int(forced.sum().item())
            keys, _, n = S.prune(out_set.keys, out_set.n, None, mask, n_keep)
        kept = S.CoordSet(keys, n, ts_out, out_set.bounds)
        # the up-sampled features, for the kept rows only: transposed conv restricted to them, in pair-list form
        if "_packed_conv" not in gen.__dict__:
            gen.__dict__["_packed_conv"] = S.PackedConv(transposed=False)
        packed = gen._packed_conv.get(gen.kernel, state_dict_order=True)
        rows_form = bool(L.load().pcc_conv_pairs_supported(125, gen.in_channels, gen.out_channels))
        total = L.counter() if rows_form else None
        csr5 = cs_in.csr_for(kept.keys, n, 5, ts_out, total=total)
        nxt = None
        if next_up is not None and n > 0:
            g2 = next_up[-1]
            if isinstance(g2, ME.MinkowskiGenerativeConvolutionTranspose) and ts_out % g2.stride == 0:
                nxt = kept.expand_begin(g2.kernel_size, ts_out // g2.stride, want_csr=False)
        pairs = S.resolve(S.Pending(total, lambda v: int(v[0])) if rows_form else None, nxt)[0]
        if rows_form:
            xk = S.convt_forward_rows(feats, packed, gen.bias, 125, gen.in_channels, gen.out_channels, csr5, n, pairs_bound=pairs)
        else:       # narrow shapes: slot map of the kept rows + the generic convolution
            xk = S.conv_forward(feats, packed, gen.bias, 125, gen.in_channels, gen.out_channels,
                                S.map_from_csr(csr5, cs_in.n, n, 5), n)
        if probe is not None:
            probe("kept", lvl, kept, None, None, xk)
        return SparseTensor._from_canonical(kept, xk), pred, mask

    def forward(self, y, coords=None, k=None, trace=None, probe=None):
        """y (stride 8) -> x (stride 1) features at the k-selected voxels (`model/transforms.py:170-225`).
        trace (dict): run layer by layer and record every level's keys / features / logits / mask.
        probe (callable, tests): probe("select", lvl, candidate set, logits [n,1], mask, feats | None) may return a
        replacement mask (the full-size parity test pins the rows inside the float-noise band around the k-th logit to
        the oracle's choice); probe("kept", lvl, kept set, None, None, kept features) observes the pruned tensor."""
        predictions = []
        x = y
        ups = (self.up_1, self.up_2, self.up_3)
        for lvl, (up, head) in enumerate(((self.up_1, self.predict_1), (self.up_2, self.predict_2),
                                          (self.up_3, self.predict_3))):
            if coords is None and trace is None and self._can_fuse(up, head, x if lvl else y):
                x, pred, _ = self._up_predict_fused(up, head, x, k[lvl], probe, lvl, ups[lvl + 1] if lvl < 2 else None)
                predictions.append(pred)
                continue
            x = up(x)
            pred = self._predict(head, x)
            mask, n_keep = self._topk_prediction(pred, k[lvl])
            if probe is not None:
                forced = probe("select", lvl, x._cset, pred.F, mask, x.F)
                if forced is not None:
                    mask, n_keep = forced, int(forced.sum().item())
            if trace is not None:
                trace[f"keys_{lvl}"], trace[f"feats_{lvl}"] = x._cset.keys[:x._cset.n], x.F
                trace[f"logit_{lvl}"], trace[f"mask_{lvl}"] = pred.F, mask
            predictions.append(pred)
            x = self._prune_tensor(x, mask, n_keep)
            if probe is not None:
                probe("kept", lvl, x._cset, None, None, x.F)
        x = self.color_conv(x)
        if coords is None:
            return x
        with torch.no_grad():   # training targets: coordinates of the ground truth at strides 2 and 4
            cs1 = coords._cset.stride(coords._cset.ts * 2)
            cs2 = cs1.stride(cs1.ts * 2)
            ones = lambda c: SparseTensor._from_canonical(c, torch.ones((c.n, 1), device=c.device))
        return x, [ones(cs2), ones(cs1), coords], predictions
