#!/usr/bin/env python3
"""Frame-parallel evaluation sweep (BASELINE configs 3 and 5): synthetic vox10-like / vox11-like frames are assigned to the
ranks (one process per GPU, longest first), each rank codes its frames to byte strings and decodes them, measures
bits / point and the D1 / colour PSNRs on its GPU, and one all_gather of fixed-size records ends the sweep
(`evaluate.py:102-195` without the external tools).  Single process: python tools/eval_frames.py [--bits 9 9 10]
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
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
model = bench.build_model(dev)
q = torch.tensor([[0.5, 0.5]], device=dev)
clouds = {}


def cloud(i):
    if i not in clouds:
        clouds[i] = synth.surface_cloud(seed=i, bits=args.bits[i])
    return clouds[i]


sizes = [int(0.75 * 4 ** b) for b in args.bits]          # surface area grows with the square of the resolution


def process(i):
    pc = torch.from_numpy(cloud(i)).to(dev)
    bs = args.block_size or (1024 if args.bits[i] <= 10 else 512)
    torch.cuda.synchronize(); t0 = time.time()
    out = model.compress(pc, q, block_size=bs)
    torch.cuda.synchronize(); t1 = time.time()
    rec = model.decompress(coordinates=[c.clone() for c in out[3]], strings=out[0], shape=out[1], k=out[2], q_vals=out[4])
    torch.cuda.synchronize(); t2 = time.time()
    rep = metrics.pointcloud_metrics(pc, rec, resolution=(1 << args.bits[i]) - 1)
    extra[i] = (rep["sym_psnr_mse"], rep["sym_y_psnr"], len(out[0]))
    return (i, pc.shape[0], t1 - t0, t2 - t1, metrics.count_bits(out[0]), rec.shape[0])


extra = {}
for i in frames.assign(sizes, world)[rank][:1]:
    process(i)                                         # warm-up on this rank's first frame (allocator, weight packing)
extra.clear()
if world > 1:
    dist.barrier(device_ids=[local])
torch.cuda.synchronize()
t_wall0 = time.time()
recs = frames.run_sharded(sizes, process, dev, rank, world)
torch.cuda.synchronize()
wall = torch.tensor([time.time() - t_wall0], dtype=torch.float64, device=dev)
if world > 1:
    dist.all_reduce(wall, op=dist.ReduceOp.MAX)        # the sweep ends when the slowest rank ends
wall = float(wall.item())
if rank == 0:
    print(f"{'frame':>5s} {'points':>9s} {'enc ms':>8s} {'dec ms':>8s} {'bpp':>7s} {'decoded':>9s}")
    for r in recs:
        print(f"{int(r[0]):5d} {int(r[1]):9d} {r[2] * 1e3:8.1f} {r[3] * 1e3:8.1f} {r[4] / r[1]:7.3f} {int(r[5]):9d}")
    tot = sum(r[1] for r in recs)
    print(f"{len(recs)} frames, {tot} points in {wall:.3f} s wall (max over ranks, includes the D1 / colour report of every "
          f"frame) = {tot / wall / 1e6:.2f} M points/s over {world} rank(s); coding time alone, perfectly balanced: "
          f"{tot / sum(r[2] + r[3] for r in recs) * world / 1e6:.1f} M points/s")
for i, (d1, y, nb) in sorted(extra.items()):
    print(f"rank {rank} frame {i}: blocks {nb}  sym D1-PSNR {d1:.2f} dB  sym Y-PSNR {y:.2f} dB", flush=True)
if world > 1:
    dist.barrier(device_ids=[local])
    dist.destroy_process_group()
