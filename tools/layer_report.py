#!/usr/bin/env python3
"""Per-layer report of one encode+decode step: rows, pairs, algorithmic FLOPs, time, TFLOP/s (diagnostic tool)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from unified_point_cloud_compression_amd import sparse as S, synth  # noqa: E402

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
model = bench.build_model(dev, coder="symbols")
pc = torch.from_numpy(synth.surface_cloud(0, bits)).to(dev)
q = torch.tensor([[0.5, 0.5]], device=dev)
bench.step(model, pc, q)
calls = []
orig = S.conv_forward


def spy(feats, packed_w, bias, K, cin, cout, kmap, n_out, act=0, slope=0.01):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(feats, packed_w, bias, K, cin, cout, kmap, n_out, act, slope)
    e1.record()
    calls.append((kmap, K, cin, cout, feats.shape[0], n_out, e0, e1))
    return out


orig_t = S.convt_forward


def spy_t(feats, packed_w, bias, K, cin, cout, kmap, n_out, act=0, slope=0.01):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig_t(feats, packed_w, bias, K, cin, cout, kmap, n_out, act, slope)
    e1.record()
    calls.append((kmap, -K, cin, cout, feats.shape[0], n_out, e0, e1))
    return out


orig_c = S.convt_forward_csr


def spy_c(feats, packed_w, bias, K, cin, cout, csr, n_out, act=0, slope=0.01, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig_c(feats, packed_w, bias, K, cin, cout, csr, n_out, act, slope, **kw)
    e1.record()
    calls.append((int(csr[0][n_out].item()), -K, cin, cout, feats.shape[0], n_out, e0, e1))
    return out


orig_r = S.convt_forward_rows


def spy_r(feats, packed_w, bias, K, cin, cout, csr, n_out, act=0, slope=0.01, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig_r(feats, packed_w, bias, K, cin, cout, csr, n_out, act, slope, **kw)
    e1.record()
    calls.append((int(csr[0][n_out].item()), -K, cin, cout, feats.shape[0], n_out, e0, e1))
    return out


orig_h = S.conv_head_forward


def spy_h(feats, packed_w0, bias0, cmid, w2, bias2, cset, kmap):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig_h(feats, packed_w0, bias0, cmid, w2, bias2, cset, kmap)
    e1.record()
    calls.append((kmap, 27, feats.shape[1], cmid, feats.shape[0], feats.shape[0], e0, e1))     # + the cmid -> 1 projection and gather
    return out


orig_g = S.convt_forward_csr_grid


def spy_g(feats, packed_w, bias, K, cin, cout, csr, out_set, act, ex_bias, slope=0.01):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig_g(feats, packed_w, bias, K, cin, cout, csr, out_set, act, ex_bias, slope)
    e1.record()
    calls.append((S.csr_pair_total(csr, out_set.n), -K, cin, cout, feats.shape[0], out_set.n, e0, e1))
    return out


S.convt_forward_csr_grid = spy_g
S.COUNT_PAIRS, S.conv_forward, S.convt_forward, S.convt_forward_csr, S.convt_forward_rows = True, spy, spy_t, spy_c, spy_r
S.conv_head_forward = spy_h

bench.step(model, pc, q)
torch.cuda.synchronize()
tot_ms = tot_fl = 0
print(f"{'K':>4s} {'cin':>4s} {'cout':>4s} {'n_in':>9s} {'n_out':>9s} {'pairs':>10s} {'P/n_out':>7s} {'GF':>8s} {'ms':>8s} {'TF/s':>7s}")
for kmap, K, cin, cout, n_in, n_out, e0, e1 in calls:
    p = kmap if isinstance(kmap, int) else (kmap.pairs() if kmap is not None else n_out)
    ms = e0.elapsed_time(e1)
    fl = 2.0 * p * cin * cout
    tot_ms += ms
    tot_fl += fl
    print(f"{K:4d} {cin:4d} {cout:4d} {n_in:9d} {n_out:9d} {p:10d} {p/max(n_out,1):7.1f} {fl/1e9:8.1f} {ms:8.3f} {fl/ms/1e9:7.1f}")
print(f"total {tot_fl/1e9:.1f} GF in {tot_ms:.2f} ms = {tot_fl/tot_ms/1e9:.1f} TF/s")
if len(sys.argv) > 2:           # call list for tools/kernel_attribution.py
    import json
    rows = []
    for kmap, K, cin, cout, n_in, n_out, e0, e1 in calls:
        p = kmap if isinstance(kmap, int) else (kmap.pairs() if kmap is not None else n_out)
        rows.append(dict(K=K, cin=cin, cout=cout, n_in=n_in, n_out=n_out, pairs=p, gflop=2.0 * p * cin * cout / 1e9,
                         call_ms=e0.elapsed_time(e1)))
    json.dump(rows, open(sys.argv[2], "w"))
