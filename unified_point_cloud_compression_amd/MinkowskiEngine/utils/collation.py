"""`ME.utils.collation.sparse_collate` (`train.py:199-201`): prepend the batch index, concatenate (A.9)."""
import torch


def batched_coordinates(coords, dtype=torch.int32, device=None):
    out = []
    for b, c in enumerate(coords):
        c = torch.as_tensor(c)
        col = torch.full((c.shape[0], 1), b, dtype=c.dtype, device=c.device)
        out.append(torch.cat([col, c], dim=1))
    bc = torch.cat(out, dim=0)
    if bc.dtype.is_floating_point:
        bc = bc.floor()
    bc = bc.to(dtype)
    return bc.to(device) if device is not None else bc


def sparse_collate(coords, feats, labels=None, dtype=torch.int32, device=None):
    bc = batched_coordinates(coords, dtype=dtype, device=device)
    f = torch.cat([torch.as_tensor(x) for x in feats], dim=0)
    if device is not None:
        f = f.to(device)
    if labels is not None:
        lab = torch.cat([torch.as_tensor(x) for x in labels], dim=0)
        return bc, f, (lab.to(device) if device is not None else lab)
    return bc, f
