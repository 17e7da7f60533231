#!/usr/bin/env python3
"""Block-parallel evaluation sweep (BASELINE configs 3 and 5): synthetic vox10-like / vox11-like frames are cut
into their blocks (`partition`), the (frame, block) items are assigned to the ranks (one process per GPU, longest first over
ALL items, so a vox11 frame's >= 8 blocks spread over the node), each rank codes its items to byte strings and decodes
them from plain coordinates, measures bits and the block-local D1 numerators on its GPU, and one all_gather of fixed-size
records ends the sweep; per-frame totals are summed from the block records (`evaluate.py:102-195` without the external tools).  Single process: python tools/eval_frames.py [--bits 9 9 10]
N GPUs:  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/eval_frames.py"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from unified_point_cloud_compression_amd import frames, metrics, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bits", type=int, nargs="+", default=[10, 10, 11, 10], help="grid bits of the synthetic frames")
ap.add_argument("--block-size", type=int, default=None, help="default: 1024 for <= 10 bits, 512 above (evaluate.py:39-46)")
args = ap.parse_args()
rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
# PCC_BENCH_REHEARSE=1: the N-rank sweep on ONE GPU (every rank on device 0, collectives over gloo on host tensors) -- a check of
# the sharded control flow where no multi-GPU node is at hand, never a measurement
rehearse = world > 1 and os.environ.get("PCC_BENCH_REHEARSE", "0") == "1"
gpu = 0 if rehearse else local
if world > 1:
    frames.pin_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))    # own cores, before the first GPU call
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if rehearse:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", gpu))
torch.cuda.set_device(gpu)
dev = torch.device("cuda", gpu)
coll = torch.device("cpu") if rehearse else dev           # where the collectives' tensors live


def barrier():
    dist.barrier() if rehearse else dist.barrier(device_ids=[gpu])


model = bench.build_model(dev)
q = torch.tensor([[0.5, 0.5]], device=dev)
clouds = {}


def cloud(i):
    if i not in clouds:
        clouds[i] = torch.from_numpy(synth.surface_cloud(seed=i, bits=args.bits[i])).to(dev)
    return clouds[i]


def block_size(i):
    return args.block_size or (1024 if args.bits[i] <= 10 else 512)


# ---- partition pass: frames dealt round-robin, every rank learns every frame's block sizes (the work list) -------------------
frame_blocks = {}
local_sizes = {}
for i in range(len(args.bits)):
    if i % world == rank:
        frame_blocks[i] = model.blocks_of(cloud(i), block_size(i))
        local_sizes[i] = [int(b.shape[0]) for b in frame_blocks[i]]
sizes = frames.gather_block_sizes(local_sizes, coll, rank, world)
balance = frames.load_balance(sizes, world)


def blocks(i):
    if i not in frame_blocks:
        frame_blocks[i] = model.blocks_of(cloud(i), block_size(i))
    return frame_blocks[i]


def process(f, b):
    """One (frame, block) item: code, decode from plain coordinates, block-local D1 numerators."""
    x = blocks(f)[b]
    torch.cuda.synchronize(); t0 = time.time()
    strings, shape, k, yc = model.compress_block(x, q)
    torch.cuda.synchronize(); t1 = time.time()
    rec = model.decompress(coordinates=[yc.clone()], strings=[strings], shape=[shape], k=[k], q_vals=[q])
    torch.cuda.synchronize(); t2 = time.time()
    rep = metrics.pointcloud_metrics(x, rec, resolution=(1 << args.bits[f]) - 1)
    return (f, b, x.shape[0], t1 - t0, t2 - t1, metrics.count_bits([strings]), rec.shape[0],
            rep["AB_mse"] * x.shape[0], x.shape[0], rep["BA_mse"] * rec.shape[0], rec.shape[0])


items = frames.block_items(sizes)
mine = frames.assign([n for _, _, n in items], world)[rank]
if mine:
    process(*items[mine[0]][:2])                       # warm-up on this rank's first item (allocator, weight packing)
if world > 1:
    barrier()
torch.cuda.synchronize()
t_wall0 = time.time()
recs, totals = frames.run_sharded_blocks(sizes, process, coll, rank, world)
torch.cuda.synchronize()
wall = torch.tensor([time.time() - t_wall0], dtype=torch.float64, device=coll)
if world > 1:
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)        # the sweep ends when the slowest rank ends
wall = float(wall.item())
if rank == 0:
    print(f"{'frame':>5s} {'blocks':>6s} {'points':>9s} {'enc ms':>8s} {'dec ms':>8s} {'bpp':>7s} {'decoded':>9s} {'D1 dB':>7s}")
    import math
    for f, t in sorted(totals.items()):
        res = (1 << args.bits[f]) - 1
        mse = max(t["mse_ab"], t["mse_ba"])
        d1 = 10 * math.log10(res * res / mse) if mse > 0 else float("inf")
        print(f"{f:5d} {t['blocks']:6d} {t['n_points']:9d} {t['t_encode'] * 1e3:8.1f} {t['t_decode'] * 1e3:8.1f} {t['bpp']:7.3f} "
              f"{t['n_decoded']:9d} {d1:7.2f}")
    tot = sum(t["n_points"] for t in totals.values())
    busy = sum(t["t_encode"] + t["t_decode"] for t in totals.values())
    print(f"{len(totals)} frames as {len(recs)} (frame, block) items over {world} rank(s): {tot} points in {wall:.3f} s wall (max over "
          f"ranks, includes the block-local D1 report) = {tot / wall / 1e6:.2f} M points/s; point load max/mean over ranks {balance:.3f}; "
          f"coding time alone, perfectly balanced: {tot / busy * world / 1e6:.1f} M points/s")
if rank == 0 and rehearse:
    print(f"REHEARSAL: {world} ranks shared one GPU over gloo -- control flow only, the rates above are not measurements")
if world > 1:
    barrier()
    dist.destroy_process_group()
