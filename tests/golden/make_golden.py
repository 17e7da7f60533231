"""Generates tests/golden/codec_seed{0,1,2}.npz from the ORACLE (numpy restatement).

The reference ships no fixtures and cannot run here (MinkowskiEngine / CompressAI absent), so these
goldens pin the oracle against regressions and give the HIP path a fixed target; they are not reference outputs.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import codec  # noqa: E402
from unified_point_cloud_compression_amd import synth  # noqa: E402

GAIN = {0: 4.0, 1: 4.0, 2: 4.0}


def case(seed):
    adaptive = seed == 2
    cfg = codec.small_config(adaptive=adaptive, offsets=adaptive, inverse=False)
    P = codec.random_params(cfg, seed, gain=GAIN[seed])
    pc = synth.random_block(seed, 32, 0.08)
    q = np.array([[0.3 + 0.2 * seed, 0.6]], dtype=np.float32)
    blocks = codec.compress(P, cfg, pc, q)
    trace = {}
    rec = codec.decompress(P, cfg, blocks, trace=trace)
    b = blocks[0]
    return dict(seed=seed, adaptive=adaptive, q=q, n_points=b["n_points"], k=np.array(b["k"], dtype=np.int64),
                y_keys=b["y_keys"], z_keys=b["z_keys"], y_symbols=b["y_symbols"], z_symbols=b["z_symbols"],
                indexes=b["indexes"], bits=np.float64(codec.bits(blocks)), recon=rec,
                mask_0=trace["mask_0"], mask_1=trace["mask_1"], mask_2=trace["mask_2"], keys_2=trace["keys_2"])


if __name__ == "__main__":
    out = os.path.dirname(os.path.abspath(__file__))
    for seed in (0, 1, 2):
        d = case(seed)
        np.savez_compressed(os.path.join(out, f"codec_seed{seed}.npz"), **d)
        print(seed, "points", d["n_points"], "k", d["k"].ravel().tolist(), "bits/pt", d["bits"] / d["n_points"],
              "sym absmax", np.abs(d["y_symbols"]).max(), np.abs(d["z_symbols"]).max(),
              "nonzero y frac", (d["y_symbols"] != 0).mean())
