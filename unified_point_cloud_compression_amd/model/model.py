"""`UnifiedModel`: counterpart of the reference's `model/model.py:15-250` (compress / decompress / update).

Same call signatures and return structure as the reference when `path is None`:
  compress(pointcloud[N,6], q[1,2], block_size) -> (bitstreams, block_shapes, block_k, block_coordinates,
  block_q_vals), one entry per block; decompress(coordinates=, strings=, shape=, k=, q_vals=) -> [N,6].
`bitstreams[i]` = [[y_string], [z_string]] rANS byte strings as in the reference (`utils.count_bits` applies).  With
`path=` the blocks go through the file container of `container.py` (field order of `model/model.py:253-385`) with the
stride-8 coordinates coded losslessly by the octree coder of libpcc_hip instead of the external `tmc3` subprocess.
"""
import torch

from .. import container
from .. import lib as L
from .. import sparse as S
from ..MinkowskiEngine.sparse_tensor import SparseTensor
from ..compressai.models.base import CompressionModel
from .entropy_models import MeanScaleHyperprior
from .transforms import AnalysisTransform, SparseSynthesisTransform


class UnifiedModel(CompressionModel):
    def __init__(self, config):
        super().__init__()
        self.g_a = AnalysisTransform(config["g_a"])
        self.g_s = SparseSynthesisTransform(config["g_s"])
        self.entropy_model = MeanScaleHyperprior(config["entropy_model"])

    PLAN_SYNTHESIS_EARLY = __import__("os").environ.get("PCC_PLAN_HS", "1") != "0"

    def update(self):
        self.entropy_model.update(force=True)

    def aux_loss(self):
        return self.entropy_model.aux_loss()

    def forward(self, x, q, Lambda):
        """Training forward (`model/model.py:45-90`): x = SparseTensor of colours; returns the dict the losses consume:
        prediction, points (ground-truth coordinates at strides 4, 2, 1), occ_predictions, q_map, likelihoods."""
        feats = torch.cat([torch.ones((x.C.shape[0], 1), device=x.device), x.F], dim=1)
        x = SparseTensor(coordinates=x.C, features=feats, device=x.device)
        chain = x._cset.stride_chain_begin(self.stride_chain())      # g_a's and h_a's output sets: one batch of launches, one read
        if chain is not None:
            S.resolve(chain)
        coords = SparseTensor._from_canonical(x._cset, torch.ones((x._cset.n, 1), device=x.device))
        y, k = self.g_a(x)
        y_hat, likelihoods = self.entropy_model(y, q)
        x_hat, points, predictions = self.g_s(y_hat, coords=coords, k=k)
        return {"prediction": x_hat, "points": points, "occ_predictions": predictions, "q_map": Lambda,
                "likelihoods": {"y": likelihoods[0], "z": likelihoods[1]}}

    @staticmethod
    def partition(pointcloud, block_size):
        """Block partition (`model/model.py:121-127`): blocks ordered by (ix, iy, iz), points keep input order."""
        xyz = pointcloud[:, :3]
        mn = xyz.t().contiguous().amin(dim=1)      # row reductions of the transposed copy: 8x faster than a column amin
        bi = ((xyz - mn) / block_size).floor().to(torch.int64)
        code = bi[:, 0] * 10 ** 6 + bi[:, 1] * 10 ** 3 + bi[:, 2]
        if int(code.max().item()) == 0:
            return None, [pointcloud.shape[0]]
        order = torch.argsort(code, stable=True)
        _, counts = torch.unique_consecutive(code[order], return_counts=True)
        return order, counts.tolist()

    def stride_chain(self):
        """Tensor strides of the sets the encoder derives from the input one after the other (g_a's three stride-2
        convolutions, then h_a's two): the input set's constructor builds them all in its own batch of launches."""
        chain = self.__dict__.get("_stride_chain")          # (the module tree does not change after construction)
        if chain is None:
            chain, ts = [], 1
            for m in list(self.g_a.modules()) + list(self.entropy_model.h_a.modules()):
                if getattr(m, "stride", 1) != 1 and hasattr(m, "kernel_size"):
                    ts *= m.stride
                    chain.append(ts)
            chain = self.__dict__["_stride_chain"] = tuple(chain)
        return chain

    def block_input(self, x_block, coords=None):
        """floor -> int32, de-duplicate (first wins), features [1, r, g, b] (`model/model.py:141-161`)."""
        n = x_block.shape[0]
        if coords is None:
            coords = torch.cat([torch.zeros((n, 1), device=x_block.device, dtype=x_block.dtype), x_block[:, :3]], dim=1)
        if not torch.is_grad_enabled():
            coords._pcc_chain = self.stride_chain()
        feats = torch.cat([torch.ones((n, 1), device=x_block.device, dtype=torch.float32),
                           x_block[:, 3:6].to(torch.float32)], dim=1)
        return SparseTensor(coordinates=coords, features=feats, device=x_block.device)

    # ---- blocks as independent work items (the shard unit across GPUs) ------------------------------------------
    def blocks_of(self, pointcloud, block_size=1024):
        """The frame's blocks in the order `compress` codes them (`partition`): a list of [n_i, 6] tensors.  Blocks carry no
        halo, so any subset may be coded on another rank (`frames.run_sharded_blocks`) and decoded there: `decompress` takes
        per-block lists of any length."""
        order, counts = self.partition(pointcloud, block_size)
        xs = pointcloud if order is None else pointcloud[order]
        out, start = [], 0
        for c in counts:
            out.append(xs[start:start + c])
            start += c
        return out

    @torch.no_grad()
    def compress_block(self, x_block, q, coords=None):
        """One block through g_a and the entropy model: (strings, shape, k, latent coordinates) -- the body of the block loop
        of `compress` (`model/model.py:137-176`).  The matrix products of g_a and h_a run in the three-term fp16 form under
        its range guard: should a layer's operands leave the range the form's error bound covers, THIS
        block is coded again with those products in the six-term 24-bit form.  The hyper-synthesis never changes form (its
        scope is pinned, `MeanScaleHyperprior._gaussian_params`): the decoder reproduces its bits whatever happened here."""
        try:
            return self._compress_block(x_block, q, coords)
        except L.RangeGuardTripped:
            with L.arith_scope(L.ARITH_BF6):
                return self._compress_block(x_block, q, coords)

    def _compress_block(self, x_block, q, coords=None):
        if isinstance(coords, tuple):                       # (FrameRows, features): the frame went through `frame_intake`
            coords[0]._pcc_chain = self.stride_chain()
            x = SparseTensor._from_frame(x_block, coords[0], coords[1])
        else:
            x = self.block_input(x_block, coords=coords)
        y, k = self.g_a(x)
        _, symbols, shape = self.entropy_model.compress(y, q)
        return symbols, shape, k, y.C

    # ---- file container (`model/model.py:253-486`) ---------------------------------------------------------
    def save_bitstream(self, path, blocks_coordinates, blocks_strings, blocks_shapes, blocks_k, blocks_q):
        return container.save_bitstream(path, blocks_coordinates, blocks_strings, blocks_shapes, blocks_k, blocks_q)

    def load_bitstream(self, path):
        return container.load_bitstream(path)

    def gpcc_encode(self, points, directory=None):
        """Lossless coding of the latent coordinates; name kept from the reference (`model/model.py:388-440`), the
        coder is libpcc_hip's octree coder, not the external G-PCC binary."""
        return container.encode_points(points)

    def gpcc_decode(self, bin, directory=None):
        return torch.from_numpy(container.decode_points(bin))

    @torch.no_grad()
    def compress(self, pointcloud, q, path=None, block_size=1024, scaling_factor=1.0):
        """`UnifiedModel.compress` (`model/model.py:94-187`); the range-guard fallback is per block (`compress_block`)."""
        if path and self.entropy_model.entropy_coder == "symbols":
            raise L.PccError("path= needs byte strings: build the model with entropy_coder 'pcc_streams' or 'ans'")
        if not pointcloud.is_cuda:
            raise L.PccError("compress expects the point cloud on the GPU (`utils.py:436-441` moves it there)")
        if scaling_factor != 1.0:
            pointcloud = pointcloud.clone()
            pointcloud[:, :3] = torch.round(pointcloud[:, :3] / scaling_factor).int()
        # one block (the usual case: `block_size` 1024 against vox10 frames)?  The bounds the coordinate set needs anyway
        # answer that: floor(max) - floor(min) < block_size on every axis implies a single block of `partition`, so its two
        # reductions and its host read are skipped and the read below also serves the SparseTensor constructor.
        single = None
        n_pts = pointcloud.shape[0]
        if n_pts > 1 and pointcloud.dtype == torch.float32 and pointcloud.dim() == 2 and pointcloud.shape[1] == 6 \
                and pointcloud.is_contiguous() and pointcloud.data_ptr() % 8 == 0:
            # keys, features [1, r, g, b], bounds and order flag in one kernel + one read (`pcc_frame_intake`)
            keys, feats, b, canonical = S.frame_intake(pointcloud)
            if max(b.hi[i] - b.lo[i] for i in range(3)) < block_size:
                single = (S.FrameRows(n_pts, pointcloud.device, (keys, b, canonical)), feats)
        elif n_pts > 1:
            c4 = torch.cat([torch.zeros((n_pts, 1), device=pointcloud.device, dtype=pointcloud.dtype), pointcloud[:, :3]], dim=1)
            keys = S.pack_keys(c4)
            b, canonical = S.bounds_of(c4, canon_keys=keys)
            if max(b.hi[i] - b.lo[i] for i in range(3)) < block_size:
                c4._pcc_hint = (keys, b, canonical)
                single = c4
        order, counts = (None, [n_pts]) if single is not None else self.partition(pointcloud, block_size)
        xs = pointcloud if order is None else pointcloud[order]
        bitstreams, block_shapes, block_coordinates, block_q_vals, block_k = [], [], [], [], []
        start = 0
        for count in counts:
            symbols, shape, k, yc = self.compress_block(xs[start:start + count], q, coords=single)
            block_q_vals.append(q)
            block_coordinates.append(yc)
            block_shapes.append(shape)
            block_k.append(k)
            bitstreams.append(symbols)
            start += count
        if path:
            self.save_bitstream(path, block_coordinates, bitstreams, block_shapes, block_k, block_q_vals)
            return None
        return bitstreams, block_shapes, block_k, block_coordinates, block_q_vals

    @torch.no_grad()
    def decompress(self, path=None, coordinates=None, strings=None, shape=None, k=None, q_vals=None, trace=None,
                   probe=None):
        """`UnifiedModel.decompress` (`model/model.py:189-250`).  The synthesis transform runs under the same range guard as
        the encoder's analysis: when it trips, g_s alone (never the entropy model: its hyper-synthesis is pinned to one form
        on both sides) is evaluated again in the six-term form on the latents already decoded."""
        device = self.g_s.down_conv.kernel.device
        if path:
            coordinates, strings, shape, k, q_vals = self.load_bitstream(path)
            for i, c in enumerate(coordinates):                 # [n,3] -> [n,4] with batch column 0 (`model/model.py:218-222`)
                c = c.to(device)
                coordinates[i] = torch.cat([torch.zeros((c.shape[0], 1), dtype=c.dtype, device=device), c], dim=1)
                q_vals[i] = q_vals[i].to(device)
        feats, coords, status, blocks, latents = [], [], [], [], []
        for i, (block_symbols, block_shape, block_coords, block_k) in enumerate(zip(strings, shape, coordinates, k)):
            # (the stride-32 set cannot have more rows than the stride-8 one: a header that says otherwise is refused before
            #  anything is sized by it)
            if not 0 <= int(block_shape[0]) <= max(int(block_coords.shape[0]), 0):
                raise L.PccError(f"bitstream says {int(block_shape[0])} hyper-latent rows for {int(block_coords.shape[0])} latent rows")
            pre = self.entropy_model.predecode(block_symbols, block_shape, device, check=status)
            y_cset = getattr(block_coords, "_pcc_cset", None)
            if (y_cset is None or y_cset.ts != 8 or y_cset.n != block_coords.shape[0]
                    or getattr(block_coords, "_pcc_version", block_coords._version) != block_coords._version):
                # plain coordinates (what a decoder gets: `utils.py:461-465`, `load_bitstream`): the canonical set of the
                # rows (the reference builds `ME.SparseTensor(coordinates=points[0], tensor_stride=8)`, `model/entropy_models.py:439`)
                y_cset = S.coordset_from_coords(block_coords.to(device), 8)[0]
            # z coordinates: two k3-s2 `down_conv`s in the reference (`model/model.py:227-229`) = coordinate-only stride;
            # floor(floor(c/16)*16/32)*32 = floor(c/32)*32, so the stride-32 set comes straight from y's rows.  Everything
            # else that depends on y's COORDINATES alone (first synthesis level: 5x5x5 pair list, candidate set) is queued
            # with it and the sizes come back in one read.
            ts_z = y_cset.ts * 4
            if y_cset.n == 0:
                z_cset = y_cset.stride(y_cset.ts * 2).stride(ts_z)
            else:
                z_cset = S.resolve(y_cset.stride_begin(ts_z), *self.g_s.plan(y_cset))[0]
            if z_cset.n != int(block_shape[0]):
                raise L.PccError(f"bitstream says {int(block_shape[0])} hyper-latent rows, the coordinates give {z_cset.n}")
            if self.PLAN_SYNTHESIS_EARLY:
                self.entropy_model.plan_synthesis(z_cset, y_cset)   # (beside the hyper-latent's decode on the side stream)
            pre = self.entropy_model.predecode_upload(pre, block_symbols, device)
            y_hat = self.entropy_model.decompress([y_cset, z_cset], block_symbols, block_shape, q_vals[i], check=status, pre=pre)
            x_hat = self.g_s(y_hat, k=block_k, trace=trace, probe=probe)
            blocks.append(x_hat)
            latents.append((y_hat, block_k))
        out = self._finish(blocks, device)
        guard = L.h_guard(device)
        flags = torch.cat([s.reshape(-1)[:1].to(torch.int32) for s in status] + [guard]).tolist()   # one deferred read: rANS containers + range guard
        if any(flags[:-1]):
            raise L.PccError("malformed rANS container in the bitstream")
        if flags[-1] and L.arith() == L.ARITH_H3:
            guard.zero_()
            with L.arith_scope(L.ARITH_BF6):
                blocks = [self.g_s(y_hat, k=block_k, trace=trace, probe=probe) for y_hat, block_k in latents]
            out = self._finish(blocks, device)
        return out

    def _finish(self, blocks, device):
        """[x, y, z, clamp(round(255 f), 0, 255) / 255] rows of the decoded blocks (`model/model.py:240-250`)."""
        out = None
        if blocks and all(x._perm is None and x.F.dim() == 2 and x.F.shape[1] == 3 and x.F.is_contiguous()
                          and x.F.dtype == torch.float32 for x in blocks):
            # [x, y, z, clamp(round(255 f), 0, 255) / 255] in one launch per block
            out = torch.empty((sum(x._cset.n for x in blocks), 6), dtype=torch.float32, device=device)
            at = 0
            for x in blocks:
                L.call("pcc_decode_finish", L.ptr(x._cset.keys), L.ptr(x.F), x._cset.n, out.data_ptr() + at * 24, L.stream())
                at += x._cset.n
        if out is not None:
            return out
        feats, coords = [], []
        for x in blocks:
            feats.append(x.F)
            coords.append(x.C)
        f = torch.cat(feats, dim=0)
        c = torch.cat(coords, dim=0)
        f = torch.clamp(torch.round(f * 255), 0.0, 255.0) / 255
        return torch.cat([c[:, 1:4].to(f.dtype), f], dim=1)
