"""CPU suite: frame sharding logic and the N>1 path (world size 2, gloo)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unified_point_cloud_compression_amd import frames
from unified_point_cloud_compression_amd.sparse import Bounds


def test_assign_balanced_and_deterministic():
    sizes = [10, 80, 30, 30, 55, 5, 70, 20]
    a = frames.assign(sizes, 3)
    assert sorted(i for r in a for i in r) == list(range(8))
    loads = [sum(sizes[i] for i in r) for r in a]
    assert max(loads) - min(loads) <= max(sizes)
    assert a == frames.assign(sizes, 3)
    assert frames.assign([], 2) == [[], []]
    assert frames.assign([7], 4) == [[0], [], [], []]


def test_bounds_bit_mask():
    b = Bounds(0, (0, 0, 0), (1023, 1023, 1023))
    assert b.bit_mask() == (0x3FF << 32) | (0x3FF << 16) | 0x3FF
    assert Bounds(3, (-2, 0, 5), (5, 0, 5)).bit_mask() == (0x3 << 48) | (0xFFFF << 32)   # sign change flips the bias bit
    assert b.strided(4).hi == (1020, 1020, 1020) and b.expanded(5, 2).lo == (-4, -4, -4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = [100, 300, 200, 50, 250]

    def process(i):                     # stand-in for compress+decompress of frame i
        return (i, sizes[i], 0.001 * sizes[i], 0.002 * sizes[i], 8.0 * sizes[i], sizes[i])

    recs = frames.run_sharded(sizes, process, torch.device("cpu"), rank, world)
    q.put((rank, recs))
    dist.barrier()
    dist.destroy_process_group()


def test_run_sharded_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0] == out[1]                                    # every rank holds all records
    assert [int(r[0]) for r in out[0]] == [0, 1, 2, 3, 4]      # ordered by frame
    assert [int(r[1]) for r in out[0]] == [100, 300, 200, 50, 250]


# ---- (frame, block) items: the real record path on 3 ranks with uneven items ------------------------------------------------
FRAME_BLOCKS = {0: [400, 350, 50], 1: [900], 2: [120, 130, 110, 140, 100, 90, 95, 105], 3: [610, 20], 4: [300, 280, 310, 290]}


def _block_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    # the partition pass: frames dealt round-robin, block sizes all-gathered (the work list)
    local = {f: s for f, s in FRAME_BLOCKS.items() if f % world == rank}
    sizes = frames.gather_block_sizes(local, dev, rank, world)
    done = []

    def process(f, b):                  # stand-in for compress + decompress + block-local D1 of block b of frame f
        n = sizes[f][b]
        done.append((f, b))
        return (f, b, n, 1e-6 * n, 2e-6 * n, 6.0 * n, n, 0.25 * n, n, 0.5 * n, n)

    recs, totals = frames.run_sharded_blocks(sizes, process, dev, rank, world)
    q.put((rank, sizes, done, recs, totals))
    dist.barrier()
    dist.destroy_process_group()


def test_block_items_sharded_over_3_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_block_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=120)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert res[r][0] == FRAME_BLOCKS                               # every rank holds the whole work list
        assert res[r][2] == res[0][2] and res[r][3] == res[0][3]       # ... and all block records / frame totals
    done = sorted(x for r in range(world) for x in res[r][1])
    assert done == [(f, b) for f, s in sorted(FRAME_BLOCKS.items()) for b in range(len(s))]     # every item coded exactly once
    assert len({r for r in range(world) if any(f == 2 for f, _ in res[r][1])}) > 1           # one frame's blocks on several ranks
    totals = res[0][3]
    for f, s in FRAME_BLOCKS.items():
        assert totals[f]["blocks"] == len(s) and totals[f]["n_points"] == sum(s) and totals[f]["n_decoded"] == sum(s)
        assert abs(totals[f]["bpp"] - 6.0) < 1e-12 and abs(totals[f]["mse_ab"] - 0.25) < 1e-12 and abs(totals[f]["mse_ba"] - 0.5) < 1e-12
    loads = [sum(FRAME_BLOCKS[f][b] for f, b in res[r][1]) for r in range(world)]
    assert max(loads) / (sum(loads) / world) <= 1.15
    assert abs(frames.load_balance(FRAME_BLOCKS, world) - max(loads) / (sum(loads) / world)) < 1e-12
    # whole frames as items would leave the ranks 1.27x apart on this list: the reason blocks are the unit
    whole = [sum(v) for _, v in sorted(FRAME_BLOCKS.items())]
    fl = [sum(whole[i] for i in r) for r in frames.assign(whole, world)]
    assert max(fl) / (sum(fl) / world) > 1.15
