#!/usr/bin/env python3
"""Experiment: does a physical Z-curve (Morton) row order speed up the big stride-1 convolutions?
Builds the decoder-sized candidate sets of the bench frame, times pcc_conv_fwd with rows in canonical (key) order and
with rows + kernel map permuted into Morton order, and checks the results are the same rows."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unified_point_cloud_compression_amd import lib as L, sparse as S, synth  # noqa: E402

dev = torch.device("cuda:0")
pc = torch.from_numpy(synth.surface_cloud(0, 10)).to(dev)
coords = torch.cat([torch.zeros(pc.shape[0], 1, device=dev, dtype=pc.dtype), pc[:, :3]], 1).int()
cs, _, _ = S.coordset_from_coords(coords, 1)
s2 = cs.stride(2)
s4 = s2.stride(4)


def part1by2(v):
    v = v & 0x3FF
    v = (v | (v << 16)) & 0x30000FF
    v = (v | (v << 8)) & 0x300F00F
    v = (v | (v << 4)) & 0x30C30C3
    v = (v | (v << 2)) & 0x9249249
    return v


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, src, ts, cin, cout in (("L2 128->64", s4, 2, 128, 64), ("L3 32->16", s2, 1, 32, 16), ("L3 32->32", s2, 1, 32, 32)):
    big = src.expand(5, ts)
    n = big.n
    kmap = big.kernel_map(big, 3)
    dense = kmap.dense()
    assert kmap.rows is None
    assert torch.equal(kmap.nbr[:27 * n].view(27, n), dense), "unexpected nbr layout"
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    pk = S.PackedConv().get(torch.nn.Parameter(w))
    x = torch.randn(n, cin, device=dev)
    t0 = timeit(lambda: S.conv_forward(x, pk, None, 27, cin, cout, kmap, n))
    ref = S.conv_forward(x, pk, None, 27, cin, cout, kmap, n)
    c = big.coords()[:, 1:].long() - torch.tensor(big.bounds.lo, device=dev)
    c = c // ts
    code = part1by2(c[:, 2]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 0]) << 2)
    perm = torch.argsort(code)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, device=dev)
    d2 = dense[:, perm].long()
    d2 = torch.where(d2 >= 0, inv[d2.clamp(min=0)], d2).int().contiguous()
    km2 = S.KernelMap()
    km2.__dict__.update(kmap.__dict__)
    km2.nbr = d2.view(-1)
    xm = x[perm].contiguous()
    t1 = timeit(lambda: S.conv_forward(xm, pk, None, 27, cin, cout, km2, n))
    out = S.conv_forward(xm, pk, None, 27, cin, cout, km2, n)
    same = torch.equal(out[inv], ref)
    pairs = int((dense >= 0).sum())
    print(f"{name}: rows {n} pairs/row {pairs / n:.1f}  key order {t0:.3f} ms  morton order {t1:.3f} ms  same={same}"
          f"  TF/s {2e-9 * pairs * cin * cout / t0:.1f} -> {2e-9 * pairs * cin * cout / t1:.1f}", flush=True)
    del big, kmap, dense, d2, x, xm, out, ref
    torch.cuda.empty_cache()
