#!/usr/bin/env python3
"""Check the z-run wave16 kernel against a torch gather reference on a decoder-like candidate set."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unified_point_cloud_compression_amd import sparse as S, synth  # noqa: E402

dev = torch.device("cuda:0")
pc = torch.from_numpy(synth.surface_cloud(0, 8)).to(dev)
coords = torch.cat([torch.zeros(pc.shape[0], 1, device=dev, dtype=pc.dtype), pc[:, :3]], 1).int()
cs, _, _ = S.coordset_from_coords(coords, 1)
s2 = cs.stride(2)
for name, big in (("surface", cs), ("blob", s2.expand(5, 1))):
    n = big.n
    kmap = big.kernel_map(big, 3)
    dense = kmap.dense().long()
    for cin, cout in ((32, 16), (16, 16), (64, 8)):
        w = torch.randn(27, cin, cout, device=dev) * 0.1
        pk = S.PackedConv().get(torch.nn.Parameter(w))
        x = torch.randn(n, cin, device=dev)
        out = S.conv_forward(x, pk, None, 27, cin, cout, kmap, n)
        ref = torch.zeros(n, cout, device=dev, dtype=torch.float64)
        for k in range(27):
            m = dense[k] >= 0
            ref[m] += x[dense[k][m]].double() @ w[k].double()
        err = (out.double() - ref).abs()
        bad = (err.amax(1) > 1e-3).nonzero().flatten()
        print(name, cin, cout, "rows", n, "max err", float(err.max()), "bad rows", bad.numel(), bad[:10].tolist(),
              (bad[:10] % 32).tolist(), flush=True)
