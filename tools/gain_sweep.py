#!/usr/bin/env python3
"""bpp of the y/z strings vs the init gain of the bench model (picks a gain giving a rate in the reference's range)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unified_point_cloud_compression_amd import synth, metrics
dev = torch.device("cuda:0")
pc = torch.from_numpy(synth.surface_cloud(0, 10)).to(dev)
q = torch.tensor([[0.5, 0.5]], device=dev)
for g in [float(a) for a in sys.argv[1:]]:
    model = bench.build_model(dev, gain=g)
    out = model.compress(pc, q)
    ys, zs = out[0][0]
    sm = bench.build_model(dev, gain=g, coder="symbols")
    o2 = sm.compress(pc, q)
    y_sym = o2[0][0][0]
    print(f"gain {g:4.2f}: bpp {metrics.count_bits(out[0])/pc.shape[0]:7.3f}  y bytes {len(ys[0])} z bytes {len(zs[0])}  nonzero y {float((y_sym!=0).float().mean()):.3f} |y|max {int(y_sym.abs().max())}", flush=True)
