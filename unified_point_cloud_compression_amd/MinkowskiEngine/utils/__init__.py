"""`ME.utils` subset the reference uses: `sparse_quantize` (`model/model.py:152-156`,
`train.py:204-208`, `data/transform.py:96-100`) and `collation.sparse_collate` (`train.py:199-201`)."""
import torch

from ... import lib as L
from ... import sparse as S
from . import collation  # noqa: F401
from .collation import sparse_collate, batched_coordinates  # noqa: F401


def sparse_quantize(coordinates, features=None, labels=None, ignore_label=-100, return_index=False,
                    return_inverse=False, quantization_size=None, device=None, **kw):
    """floor(c / quantization_size) as int32, unique rows, first occurrence kept, original order (A.9)."""
    if labels is not None or return_inverse or kw:
        raise L.PccError("sparse_quantize: labels / return_inverse are not supported")
    coords = coordinates if torch.is_tensor(coordinates) else torch.as_tensor(coordinates)
    if not coords.is_cuda:
        if device is None:
            raise L.PccError("sparse_quantize: GPU tensor (or device=) required; no CPU fallback")
        coords = coords.to(device)
    if quantization_size is not None and float(quantization_size) != 1.0:
        coords = torch.floor(coords.to(torch.float32) / float(quantization_size))
    q = coords.floor().to(torch.int32) if coords.dtype.is_floating_point else coords.to(torch.int32)
    if q.dim() != 2:
        raise L.PccError("sparse_quantize: coordinates must be 2-D")
    pad = None
    if q.shape[1] == 3:                      # unbatched coordinates: add a zero batch column for the key
        pad = torch.zeros((q.shape[0], 1), dtype=torch.int32, device=q.device)
        q4 = torch.cat([pad, q], dim=1)
    else:
        q4 = q
    cset, perm, keep = S.coordset_from_coords(q4.contiguous(), 1)
    if keep is None:
        idx = None
        out_c = q
    else:
        idx = keep
        out_c = q[keep]
    if features is None:
        return (out_c, idx if idx is not None else torch.arange(q.shape[0], device=q.device)) if return_index else out_c
    f = features if torch.is_tensor(features) else torch.as_tensor(features)
    f = f.to(q.device)
    out_f = f if idx is None else f[idx]
    if return_index:
        return out_c, out_f, (idx if idx is not None else torch.arange(q.shape[0], device=q.device))
    return out_c, out_f
