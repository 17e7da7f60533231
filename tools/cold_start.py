#!/usr/bin/env python3
"""Per-step encode / decode wall times of the first process on a box (cold-start diagnosis)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unified_point_cloud_compression_amd import synth
dev = torch.device("cuda:0")
model = bench.build_model(dev)
pc = torch.from_numpy(synth.surface_cloud(0, 10)).to(dev)
q = torch.tensor([[0.5, 0.5]], device=dev)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model.compress(pc, q, block_size=1024)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    rec = model.decompress(coordinates=out[3], strings=out[0], shape=out[1], k=out[2], q_vals=out[4])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"step {i:2d}: encode {1e3 * (t1 - t0):7.2f} ms  decode {1e3 * (t2 - t1):7.2f} ms  "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)
