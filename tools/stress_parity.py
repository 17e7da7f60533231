#!/usr/bin/env python3
"""Randomised end-to-end parity sweep (HIP path, fused inference) against the oracle: many seeds, block sizes,
densities and block partitions.  Not part of the default test run; prints one line per case and a summary."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import codec
from tests.util import load_params
from unified_point_cloud_compression_amd import synth
from unified_point_cloud_compression_amd.model import UnifiedModel
import copy

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2026)
bad = 0
t0 = time.time()
for case in range(n_cases):
    seed = int(rng.integers(0, 10 ** 6))
    size = int(rng.integers(12, 60))
    p = float(rng.uniform(0.02, 0.25))
    adaptive = bool(rng.integers(0, 2))
    block = int(rng.choice([1024, 32, 24]))
    cfg = codec.small_config(adaptive=adaptive, offsets=adaptive)
    P = codec.random_params(cfg, seed, gain=float(rng.uniform(2.0, 5.0)))
    c2 = copy.deepcopy(cfg)
    c2["entropy_model"]["entropy_coder"] = "pcc_streams"
    model = load_params(UnifiedModel(c2), P).to(dev).eval()
    model.update()
    pc = synth.random_block(seed, size, p)
    q = np.array([[float(rng.uniform(0.1, 1.0)), float(rng.uniform(0.1, 1.0))]], dtype=np.float32)
    out = model.compress(torch.from_numpy(pc).to(dev), torch.from_numpy(q).to(dev), block_size=block)
    rec = model.decompress(coordinates=out[3], strings=out[0], shape=out[1], k=out[2], q_vals=out[4]).cpu().numpy()
    ob = codec.compress(P, cfg, pc, q, block_size=block)
    ok_k = [b["k"] for b in ob] == out[2]
    ok_keys = all(np.array_equal(b["y_keys"], c._pcc_cset.keys[:c.shape[0]].cpu().numpy()) for b, c in zip(ob, out[3]))
    # decode the GPU's own strings' symbols with the oracle: take the symbols from a symbols-coder twin
    c3 = copy.deepcopy(cfg)
    c3["entropy_model"]["entropy_coder"] = "symbols"
    twin = load_params(UnifiedModel(c3), P).to(dev).eval()
    twin.update()
    so = twin.compress(torch.from_numpy(pc).to(dev), torch.from_numpy(q).to(dev), block_size=block)
    blocks = [dict(y_keys=c._pcc_cset.keys[:c.shape[0]].cpu().numpy(), y_symbols=s[0].cpu().numpy(), z_symbols=s[1].cpu().numpy(),
                   k=k, q=q) for c, s, k in zip(so[3], so[0], so[2])]
    rec_o = codec.decompress(P, cfg, blocks)
    same_n = rec.shape == rec_o.shape
    a = set(map(tuple, rec[:, :3].astype(int).tolist()))
    b = set(map(tuple, rec_o[:, :3].astype(int).tolist()))
    diff = len(a ^ b)
    status = "ok" if (ok_k and ok_keys and same_n and diff == 0) else ("near-tie" if (ok_k and ok_keys and same_n and diff <= 4) else "MISMATCH")
    bad += status == "MISMATCH"
    print(f"case {case:2d} seed {seed:6d} size {size:2d} p {p:.2f} block {block:4d} adaptive {int(adaptive)} points {pc.shape[0]:6d} "
          f"blocks {len(out[0]):2d}: k {ok_k} keys {ok_keys} voxels differing {diff} -> {status}", flush=True)
print(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
