"""`ME.SparseTensor` counterpart.

Reference constructor call sites: `model/model.py:66-70,147-161,227`, `model/blocks.py:53-56`,
`model/entropy_models.py:38-43,287-290,335-338,374-377,439-442,486-489`, `utils.py:159-164`.

Rows of `.C` / `.F` keep the order the caller supplied (first occurrence wins for duplicates);
internally every tensor also has a canonical (ascending key) view that all kernels work on.  A
tensor built from another tensor's `.C` re-uses its coordinate set and cached kernel maps in O(1)
instead of re-hashing, which is what MinkowskiEngine does on each construction.
"""
import torch

from .. import lib as L
from .. import sparse as S


def _as_stride(ts):
    if isinstance(ts, (list, tuple)):
        if len(set(int(v) for v in ts)) != 1:
            raise L.PccError(f"anisotropic tensor_stride {ts} is not supported")
        return int(ts[0])
    if torch.is_tensor(ts):
        return _as_stride(ts.tolist())
    return int(ts)


class SparseTensor:
    def __init__(self, features=None, coordinates=None, tensor_stride=1, device=None, coordinate_manager=None,
                 quantization_mode=None, **kw):
        if features is None or coordinates is None:
            raise L.PccError("SparseTensor needs features= and coordinates=")
        if kw:
            raise L.PccError(f"SparseTensor: unsupported arguments {sorted(kw)}")
        ts = _as_stride(tensor_stride)
        dev = torch.device(device) if device is not None else (
            features.device if features.is_cuda else coordinates.device)
        if dev.type != "cuda":
            raise L.PccError("this MinkowskiEngine surface is GPU-only (libpcc_hip has no CPU fallback); "
                             "pass device='cuda' or GPU tensors")
        feats = features.to(device=dev, dtype=torch.float32)
        cset = getattr(coordinates, "_pcc_cset", None)
        if cset is not None and cset.ts == ts and coordinates.device == dev and cset.n == coordinates.shape[0] \
                and getattr(coordinates, "_pcc_version", coordinates._version) == coordinates._version:
            perm = coordinates._pcc_perm
            C = coordinates
        else:
            coords = coordinates.to(dev)
            if coords.dim() != 2 or coords.shape[1] != 4:
                raise L.PccError("coordinates must be [N, 1+3] (batch index first)")
            cset, perm, keep = S.coordset_from_coords(coords, ts)
            if keep is not None:                     # duplicates: first occurrence wins (A.1)
                coords, feats = coords[keep], feats[keep]
            if perm is None:
                C = cset.coords()
            else:
                C = coords.floor().to(torch.int32) if coords.dtype.is_floating_point else coords.to(torch.int32)
                C = C.contiguous()
                C._pcc_cset, C._pcc_perm = cset, perm
        if feats.shape[0] != C.shape[0]:
            raise L.PccError(f"features have {feats.shape[0]} rows, coordinates {C.shape[0]}")
        C._pcc_version = C._version
        self._cset, self._perm, self._C, self._F = cset, perm, C, feats
        self._Fc = None

    # ---- internal -------------------------------------------------------------------------------
    @classmethod
    def _from_canonical(cls, cset, feats):
        t = cls.__new__(cls)
        t._cset, t._perm, t._F, t._Fc = cset, None, feats, None
        t._C = None
        return t

    @classmethod
    def _from_frame(cls, pc, rows, feats):
        """The tensor `SparseTensor(coordinates=[0, xyz], features=feats)` of a frame `pc` [n, 6] whose keys, bounds and
        order flag are already known (`rows`: a `sparse.FrameRows`): same coordinate set, same row order (the caller's,
        first duplicate wins), but `.C` is only materialised if somebody reads it (nothing on the compress path does)."""
        cset, perm, keep = S.coordset_from_coords(rows, 1)
        if keep is not None:
            feats = feats[keep]
        t = cls.__new__(cls)
        t._cset, t._perm, t._F, t._Fc = cset, perm, feats, None
        t._C, t._C_src = None, (pc, keep)
        return t

    def _like(self, feats_canonical):
        """Same coordinates / row order, new canonical-order features (order-preserving ops)."""
        if self._perm is None:
            return SparseTensor._from_canonical(self._cset, feats_canonical)
        t = SparseTensor.__new__(SparseTensor)
        user = torch.empty_like(feats_canonical)
        user[self._perm] = feats_canonical
        t._cset, t._perm, t._C, t._F, t._Fc = self._cset, self._perm, self._C, user, feats_canonical
        if self._C is None:
            t._C_src = getattr(self, "_C_src", None)
        return t

    def _canonical_features(self):
        if self._perm is None:
            return self._F
        if self._Fc is None:
            f = self._F
            if f.requires_grad or not f.is_cuda or f.dtype != torch.float32:
                self._Fc = f[self._perm]                 # training / odd dtypes: through autograd's index op
            else:
                f = f.contiguous()
                self._Fc = torch.empty((self._perm.shape[0], f.shape[1]), dtype=torch.float32, device=f.device)
                L.call("pcc_rows_gather", L.ptr(f), L.ptr(self._perm.contiguous()), self._perm.shape[0], f.shape[1],
                       L.ptr(self._Fc), L.stream())
        return self._Fc

    # ---- ME surface -----------------------------------------------------------------------------
    @property
    def C(self):
        if self._C is None:
            src = getattr(self, "_C_src", None)
            if self._perm is not None and src is not None:      # user-ordered rows of a frame (`_from_frame`)
                pc, keep = src
                xyz = pc[:, :3] if keep is None else pc[keep, :3]
                C = torch.cat([torch.zeros((xyz.shape[0], 1), dtype=torch.int32, device=xyz.device),
                               xyz.floor().to(torch.int32)], dim=1).contiguous()
                C._pcc_cset, C._pcc_perm = self._cset, self._perm
                self._C = C
            else:
                self._C = self._cset.coords()
            self._C._pcc_version = self._C._version
        return self._C

    coordinates = C

    @property
    def F(self):
        return self._F

    features = F

    @property
    def tensor_stride(self):
        return [self._cset.ts] * 3

    @property
    def device(self):
        return self._F.device

    @property
    def D(self):
        return 3

    @property
    def shape(self):
        return self._F.shape

    def size(self, *a):
        return self._F.size(*a)

    def __len__(self):
        return self._F.shape[0]

    def __repr__(self):
        return f"SparseTensor(n={self._F.shape[0]}, c={self._F.shape[1]}, tensor_stride={self._cset.ts}, device={self.device})"

    def features_at_coordinates(self, query_coordinates):
        """N-linear interpolation restricted to what the reference uses: queries on the tensor's own grid
        (`model/entropy_models.py:294,381,446`) -> the row at the query, zeros if absent (A.8)."""
        q = query_coordinates.to(self.device)
        qk = S.pack_keys(q)
        return S.lookup_gather(self._cset, self._canonical_features(), qk, q.shape[0])
