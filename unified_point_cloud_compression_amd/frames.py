"""Frame / block sharding across the GPUs of one node.

The reference codes frames (`evaluate.py:102-133`) and, inside a frame, spatial blocks (`model/model.py:137-176,
225-238`) fully independently -- no halo, no exchange step.  So the units are sharded statically over the ranks
(one process per GPU) with NO data-path collective; the only communication is one RCCL all_gather of fixed-size
result records at the end (a few dozen bytes per frame, xGMI bandwidth irrelevant).
"""
import torch
import torch.distributed as dist

RECORD_FIELDS = ("frame", "n_points", "t_encode", "t_decode", "bits", "n_decoded")


def assign(sizes, world_size):
    """Static longest-first greedy assignment of work items to ranks.  Returns a list (per rank) of item indices.
    Deterministic on every rank, so no scatter of the work list is needed."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda j: (load[j], j))
        out[r].append(i)
        load[r] += int(sizes[i])
    return out


def gather_records(records, device, group=None):
    """All-gather per-frame result records (list of tuples of RECORD_FIELDS floats) to every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = torch.tensor(records, dtype=torch.float64, device=device).reshape(-1, len(RECORD_FIELDS))
    if world == 1:
        return sorted(mine.cpu().tolist(), key=lambda r: r[0])
    n_local = torch.tensor([mine.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    cap = max(int(c.item()) for c in counts)
    pad = torch.zeros((cap, len(RECORD_FIELDS)), dtype=torch.float64, device=device)
    pad[:mine.shape[0]] = mine
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    out = []
    for c, b in zip(counts, bufs):
        out.extend(b[:int(c.item())].cpu().tolist())
    out.sort(key=lambda r: r[0])
    return out


def run_sharded(sizes, process, device, rank=0, world_size=1, group=None):
    """Every rank processes its share of the items with `process(index) -> record tuple`, then all ranks receive
    all records (ordered by frame index)."""
    mine = assign(sizes, world_size)[rank]
    recs = [tuple(float(v) for v in process(i)) for i in mine]
    return gather_records(recs, device, group)
