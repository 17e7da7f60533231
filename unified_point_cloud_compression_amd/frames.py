"""Frame / block sharding across the GPUs of one node.

The reference codes frames (`evaluate.py:102-133`) and, inside a frame, spatial blocks (`model/model.py:137-176,
225-238`) fully independently -- no halo, no exchange step.  So the units are sharded statically over the ranks
(one process per GPU) with NO data-path collective; the only communication is one RCCL all_gather of fixed-size
result records at the end (a few dozen bytes per frame, xGMI bandwidth irrelevant).
"""
import torch
import torch.distributed as dist

RECORD_FIELDS = ("frame", "n_points", "t_encode", "t_decode", "bits", "n_decoded")


def pin_rank(local_rank, local_world, max_threads=4):
    """Give this rank its own share of the host cores BEFORE its first GPU call.  The codec step is host-sensitive (~90 library
    calls of ~17 us of Python each): eight ranks migrating over the same cores, or eight OpenMP pools of
    `nproc` threads each, stretch every launch gap.  The cores this process may run on are cut into `local_world` contiguous
    shares; torch's intra-op pool is capped at min(share, max_threads).  Returns the list of cores (empty when the platform has
    no affinity call or the share would be empty: then nothing is changed)."""
    import os
    if not hasattr(os, "sched_getaffinity") or local_world < 1:
        return []
    cpus = sorted(os.sched_getaffinity(0))
    share = len(cpus) // local_world
    if share < 1:
        return []
    mine = cpus[local_rank * share:(local_rank + 1) * share]
    os.sched_setaffinity(0, mine)
    torch.set_num_threads(max(1, min(share, max_threads)))
    return mine


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


def gather_records(records, device, group=None, n_fields=len(RECORD_FIELDS)):
    """All-gather fixed-size result records (list of tuples of n_fields floats) to every rank, ordered by their leading
    fields (frame index, then block index for block records)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = torch.tensor(records, dtype=torch.float64, device=device).reshape(-1, n_fields)
    if world == 1:
        return sorted(mine.cpu().tolist())
    n_local = torch.tensor([mine.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    cap = max(int(c.item()) for c in counts)
    pad = torch.zeros((cap, n_fields), dtype=torch.float64, device=device)
    pad[:mine.shape[0]] = mine
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    out = []
    for c, b in zip(counts, bufs):
        out.extend(b[:int(c.item())].cpu().tolist())
    out.sort()
    return out


# ---- (frame, block) work items: the shard unit of SURVEY 8e ---------------------------------------------------------------
# A vox11 frame is >= 8 blocks at block_size 512 (`evaluate.py:39-46`) and about four vox10 frames of work: sharding whole
# frames would pin one rank on it while the others idle.  Blocks carry no halo (`model/model.py:137-176,225-238`), so the
# items of ONE frame may be coded on different ranks; their records are gathered like frame records and summed per frame.
BLOCK_FIELDS = ("frame", "block", "n_points", "t_encode", "t_decode", "bits", "n_decoded", "se_ab", "n_ab", "se_ba", "n_ba")


def gather_block_sizes(local, device, rank=0, world_size=1, group=None):
    """`local`: {frame index: [points per block]} for the frames THIS rank partitioned (frames are dealt round-robin for the
    partition pass, which is one sort per frame).  Returns the complete {frame: [sizes]} on every rank: one small all_gather
    of the work list -- the only communication before the sweep."""
    if world_size == 1 or not dist.is_initialized():
        return dict(local)
    rows = [(f, b, n) for f, sizes in sorted(local.items()) for b, n in enumerate(sizes)]
    recs = gather_records(rows, device, group, n_fields=3)
    out = {}
    for f, b, n in recs:
        out.setdefault(int(f), []).append(int(n))
    return out


def block_items(frame_block_sizes):
    """[(frame, block, points)] of {frame: [points per block]}, in (frame, block) order."""
    return [(f, b, int(n)) for f, sizes in sorted(frame_block_sizes.items()) for b, n in enumerate(sizes)]


def run_sharded_blocks(frame_block_sizes, process, device, rank=0, world_size=1, group=None):
    """Every rank codes its share of the (frame, block) items -- static longest-first assignment over ALL items, computed
    identically on every rank -- with `process(frame, block) -> BLOCK_FIELDS tuple`, then all ranks receive all block
    records.  Returns (block records, per-frame totals of `frame_totals`)."""
    items = block_items(frame_block_sizes)
    mine = assign([n for _, _, n in items], world_size)[rank]
    recs, failure = [], None
    for i in mine:
        try:
            recs.append(tuple(float(v) for v in process(items[i][0], items[i][1])))
        except Exception as e:          # a rank that dies before the collective leaves the others hanging in it: keep going to the
            failure = failure or e      # all_gather, mark the item (n_points = -1), and fail on EVERY rank afterwards
            recs.append((float(items[i][0]), float(items[i][1]), -1.0) + (0.0,) * (len(BLOCK_FIELDS) - 3))
    allr = gather_records(recs, device, group, n_fields=len(BLOCK_FIELDS))
    bad = [(int(r[0]), int(r[1])) for r in allr if r[2] < 0]
    if bad:
        raise RuntimeError(f"(frame, block) items {bad} failed on their rank") from failure
    return allr, frame_totals(allr)


def frame_totals(block_records):
    """Per frame: points, bits, decoded points, summed coding times and the D1 mean squared errors of both directions
    from the blocks' numerators (block-local nearest-neighbour association: blocks are coded without halo, so a block's
    reconstruction lies in the block's own region).  {frame: dict}."""
    out = {}
    for r in block_records:
        d = dict(zip(BLOCK_FIELDS, r))
        t = out.setdefault(int(d["frame"]), {"blocks": 0, "n_points": 0, "bits": 0.0, "n_decoded": 0, "t_encode": 0.0, "t_decode": 0.0,
                                             "se_ab": 0.0, "n_ab": 0.0, "se_ba": 0.0, "n_ba": 0.0})
        t["blocks"] += 1
        for k in ("n_points", "n_decoded"):
            t[k] += int(d[k])
        for k in ("bits", "t_encode", "t_decode", "se_ab", "n_ab", "se_ba", "n_ba"):
            t[k] += d[k]
    for t in out.values():
        t["bpp"] = t["bits"] / max(t["n_points"], 1)
        t["mse_ab"] = t["se_ab"] / max(t["n_ab"], 1.0)
        t["mse_ba"] = t["se_ba"] / max(t["n_ba"], 1.0)
    return out


def load_balance(frame_block_sizes, world_size):
    """max / mean of the per-rank point loads of the block assignment (1.0 = perfect)."""
    items = block_items(frame_block_sizes)
    loads = [sum(items[i][2] for i in r) for r in assign([n for _, _, n in items], world_size)]
    return max(loads) / (sum(loads) / world_size) if sum(loads) else 1.0


def run_sharded(sizes, process, device, rank=0, world_size=1, group=None):
    """Every rank processes its share of the items with `process(index) -> record tuple`, then all ranks receive
    all records (ordered by frame index)."""
    mine = assign(sizes, world_size)[rank]
    recs = [tuple(float(v) for v in process(i)) for i in mine]
    return gather_records(recs, device, group)
